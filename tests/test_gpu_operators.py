"""GPU: HIP operators (through the C ABI) vs the golden vectors from the reference and vs the oracle."""
import numpy as np
import pytest

from conftest import golden_cases, load_golden, rel_l2
from ksfd_amd.config import ProblemConfig
from ksfd_amd.layout import cijk_to_soa, cijk_to_petsc, cijk_to_hdf5, PETSC, SOA, HDF5
from ksfd_amd import lib as klib
from oracle import ko

pytestmark = pytest.mark.gpu
TOL = 1e-12      # rel-L2 fp64; rounding-order differences only (oracle itself is within 1e-14 of the reference)


@pytest.mark.parametrize('fused', [1, 0])
@pytest.mark.parametrize('name', golden_cases('op_'))
def test_operators_vs_reference_golden(name, fused):
    z = load_golden(name)
    cfg = ProblemConfig.from_golden(z)
    k = klib.KSFDHip(cfg)
    k.set_tuning(use_fused=fused, yseg=8)
    u = cijk_to_soa(z['u'])
    assert rel_l2(k.rhs(u), cijk_to_soa(z['rhs'])) < TOL
    assert rel_l2(k.velocity(u), cijk_to_soa(z['vel'])) < TOL
    assert rel_l2(k.jvp(cijk_to_soa(z['v']), u), cijk_to_soa(z['Jv'])) < TOL
    # groom-active input: negatives, sub-floor values, NaNs (KSFD/ksfdsym.py:888-900)
    ug = cijk_to_soa(z['ug'])
    r = k.rhs(ug)
    assert np.isfinite(r).all()
    assert rel_l2(r, cijk_to_soa(z['rhs_g'])) < TOL
    assert rel_l2(k.velocity(ug), cijk_to_soa(z['vel_g'])) < TOL
    k.close()


@pytest.mark.parametrize('name', ['op_2d_n2_aniso', 'op_3d_n1', 'op_1d_n1'])
def test_layouts_roundtrip_and_petsc_layout_rhs(name):
    """the reference's Vec is dof-fastest (KSFD/ksfdgrid.py:9-58); HDF5 datasets are (c,x,y,z) C order"""
    z = load_golden(name)
    cfg = ProblemConfig.from_golden(z)
    k = klib.KSFDHip(cfg)
    u = z['u']
    k.set_state(cijk_to_petsc(u), PETSC)
    assert np.array_equal(k.get_state(SOA), cijk_to_soa(u))
    assert np.array_equal(k.get_state(HDF5), cijk_to_hdf5(u))
    assert np.array_equal(k.get_state(PETSC), cijk_to_petsc(u))
    k.set_state(cijk_to_hdf5(u), HDF5)
    assert np.array_equal(k.get_state(SOA), cijk_to_soa(u))
    assert rel_l2(k.rhs(cijk_to_petsc(u), layout=PETSC), cijk_to_petsc(z['rhs'])) < TOL
    k.close()


@pytest.mark.parametrize('shape,nlig', [((64, 48), 1), ((250, 36), 2), ((130, 20), 3), ((16, 16, 12), 2), ((37, 21), 1), ((96,), 1)])
def test_operators_vs_oracle_seeded(shape, nlig):
    """sizes that exercise several wave strips / row segments, odd extents (generic path), 3 ligands"""
    dim = len(shape)
    if nlig <= 2:
        cfg = ProblemConfig.standard(dim, shape, L=[0.3 * (a + 1) for a in range(dim)], nlig=nlig)
    else:
        cfg = ProblemConfig(dim=dim, n=shape, L=(1.0, 0.7), lig_group=[0, 1, 0], lig_w=[1.0, 1.0, 0.5],
                            lig_s=[0.01, 0.001, 0.02], lig_gamma=[0.01, 0.001, 0.03], lig_D=[1e-6, 1e-5, 2e-6],
                            grp_alpha=[1500.0, 1500.0], grp_beta=[5.56e-4, -5.56e-4])
    rng = np.random.default_rng(5)
    N = int(np.prod(shape))
    u = 9000 + 900 * rng.standard_normal(cfg.F * N)
    v = rng.standard_normal(cfg.F * N)
    o = ko.Oracle(cfg)
    k = klib.KSFDHip(cfg)
    for yseg in (32, 5):
        k.set_tuning(use_fused=1, yseg=yseg)
        assert rel_l2(k.rhs(u), o.rhs(u)) < TOL
        assert rel_l2(k.jvp(v, u), o.jvp(u, v)) < TOL
    src = [rng.standard_normal(N) for _ in range(cfg.F)]
    for c in range(cfg.F):
        k.set_source(c, src[c])
    assert rel_l2(k.rhs(u), o.rhs(u, src)) < TOL
    vmax_o, hcfl = o.cfl(u)
    k.set_state(u)
    assert np.allclose(k.velocity_max()[:dim], vmax_o[:dim], rtol=1e-11)
    k.close()


def test_outer_loop_helpers():
    cfg = ProblemConfig.standard(2, (32, 24))
    rng = np.random.default_rng(3)
    N = 32 * 24
    u = 9000 + 900 * rng.standard_normal(2 * N)
    u[5] = np.nan
    u[7] = -3.0
    u[N + 9] = 1e-12
    k = klib.KSFDHip(cfg)
    k.set_state(u)
    k.groom()
    g = k.get_state()
    assert np.array_equal(g, ko.Oracle(cfg).groom(u))
    assert abs(k.count_worms() - g[:N].sum()) <= 1e-9 * g[:N].sum()
    k.scale_rho(0.5)
    f = np.exp(0.1 * rng.standard_normal(N))
    k.mul_rho(f)
    g2 = k.get_state()
    assert np.allclose(g2[:N], g[:N] * 0.5 * f, rtol=1e-15)
    assert np.array_equal(g2[N:], g[N:])
    k.close()


@pytest.mark.parametrize('shape,nlig', [((64, 48), 1), ((250, 36), 2), ((16, 16, 12), 2), ((132, 10, 37), 1),
                                         ((24, 9, 40), 3), ((37, 21), 1), ((96,), 1)])
def test_frozen_coefficient_jacobian_action_vs_oracle(shape, nlig):
    """u=None: the stepper's own path -- coefficient planes of the stored state once (k_jcoef), then the
    frozen-coefficient kernels (2-D strips, 3-D z-marching, generic)."""
    dim = len(shape)
    if nlig <= 2:
        cfg = ProblemConfig.standard(dim, shape, L=[0.3 * (a + 1) for a in range(dim)], nlig=nlig)
    else:
        cfg = ProblemConfig(dim=dim, n=shape, L=(1.0, 0.7, 0.9)[:dim], lig_group=[0, 1, 0], lig_w=[1.0, 1.0, 0.5],
                            lig_s=[0.01, 0.001, 0.02], lig_gamma=[0.01, 0.001, 0.03], lig_D=[1e-6, 1e-5, 2e-6],
                            grp_alpha=[1500.0, 1500.0], grp_beta=[5.56e-4, -5.56e-4])
    rng = np.random.default_rng(8)
    N = int(np.prod(shape))
    u = 9000 + 900 * rng.standard_normal(cfg.F * N)
    u[3] = np.nan
    u[N + 5] = -1.0
    v = rng.standard_normal(cfg.F * N)
    want = ko.Oracle(cfg).jvp(u, v)
    k = klib.KSFDHip(cfg)
    k.set_state(u)
    for fused in (1, 0):
        k.set_tuning(use_fused=fused)
        assert rel_l2(k.jvp(v), want) < TOL
    k.close()


@pytest.mark.parametrize('name', golden_cases('op_'))
def test_assembled_jacobian_export_vs_reference_entries(name):
    """ksfd_jacobian_csr vs every entry field the reference assembles (KSFD/ksfdsym.py:675-761, 1067-1127, 630-673),
    vs the oracle's CSR entry by entry, and CSR @ v vs the matrix-free action"""
    import scipy.sparse as sp
    from test_oracle_golden import golden_entry_fields, csr_entry_fields
    z = load_golden(name)
    cfg = ProblemConfig.from_golden(z)
    u = cijk_to_soa(z['u'])
    k = klib.KSFDHip(cfg)
    k.set_state(u)
    rowptr, col, val = k.jacobian_csr()
    orow, ocol, oval = ko.Oracle(cfg).jacobian_csr(u)
    assert np.array_equal(rowptr, orow) and np.array_equal(col, ocol)
    assert np.abs(val - oval).max() <= 1e-12 * np.abs(oval).max()
    ref, got = golden_entry_fields(z, cfg), csr_entry_fields(cfg, rowptr, col, val)
    scale = max(np.abs(f).max() for f in ref.values())
    for key, f in ref.items():
        g = got.get(key, np.zeros_like(f))
        assert np.abs(g - f).max() <= 1e-11 * max(np.abs(f).max(), 1e-300) + 1e-14 * scale, key
    F, N = cfg.F, cfg.N
    A = sp.csr_matrix((val, col, rowptr), shape=(F * N, F * N))
    v = cijk_to_soa(z['v']).reshape(F, N)
    Jv = (A @ v.T.reshape(-1)).reshape(N, F).T.reshape(-1)
    assert rel_l2(Jv, cijk_to_soa(z['Jv'])) < TOL
    assert rel_l2(Jv, k.jvp(cijk_to_soa(z['v']))) < TOL
    k.close()


def test_assembled_jacobian_direct_solve_reproduces_the_reference_style_step():
    """the exported matrix in a sparse LU (what the reference hands to MUMPS) gives the same RA34PW2 stage solve the
    matrix-free GMRES path computes: (shift I - J) y = f(u)"""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    cfg = ProblemConfig.standard(2, (48, 40), L=(0.12, 0.1), nlig=2)
    rng = np.random.default_rng(21)
    N = cfg.N
    rho = 9000 + 90 * rng.standard_normal(N)
    u = np.concatenate([rho] + [rho * cfg.lig_s[l] / cfg.lig_gamma[l] for l in range(2)])
    k = klib.KSFDHip(cfg)
    k.set_state(u)
    rowptr, col, val = k.jacobian_csr()
    F = cfg.F
    A = sp.csr_matrix((val, col, rowptr), shape=(F * N, F * N)).tocsc()
    h = 0.5
    shift = 1.0 / (0.43586652150845900 * h)
    f = k.rhs(u)
    to_petsc = lambda x: x.reshape(F, N).T.reshape(-1)
    y = spla.splu((shift * sp.identity(F * N, format='csc') - A).tocsc()).solve(to_petsc(f))
    # matrix-free: residual of the LU solution under the HIP Jacobian action
    ysoa = y.reshape(N, F).T.reshape(-1)
    r = shift * ysoa - k.jvp(ysoa) - f
    assert np.linalg.norm(r) <= 1e-10 * np.linalg.norm(f)
    k.close()


@pytest.mark.parametrize('shape,nlig,coarse', [((64, 48), 1, None), ((36, 20), 2, (5, 7)), ((16, 12, 20), 1, None), ((96,), 1, (10,)),
                                               ((37, 21), 1, (37, 4))])
def test_device_start_values_vs_oracle_and_host_generator(shape, nlig, coarse):
    """ksfd_set_state_random (row f3) vs the oracle's ko_random_function and vs ksfd_amd.initial.start_values"""
    from ksfd_amd.initial import start_values, reference_rng
    dim = len(shape)
    cfg = ProblemConfig.standard(dim, shape, L=[0.3] * dim, nlig=nlig)
    coarse = tuple(coarse) if coarse else tuple(max(1, s // 4) for s in shape)
    z = reference_rng().normal(size=coarse) * 90.0
    k = klib.KSFDHip(cfg)
    k.set_state_random(z, rho0=9000.0)
    got = k.get_state()
    N = cfg.N
    nc = list(coarse) + [1] * (3 - dim)
    want_rho = 9000.0 + ko.Oracle(cfg).random_function(nc, z.ravel(order='F'))
    assert np.abs(got[:N] - want_rho).max() <= 1e-13 * 9000.0
    for l in range(nlig):
        assert np.allclose(got[(l + 1) * N:(l + 2) * N], got[:N] * (cfg.lig_s[l] / cfg.lig_gamma[l]), rtol=1e-15, atol=0)
    host = start_values(cfg, coarse=coarse)
    assert rel_l2(got, host) < 1e-14
    k.close()


@pytest.mark.parametrize('name', golden_cases('randfn_'))
def test_device_start_values_vs_reference_random_function(name):
    """ksfd_set_state_random against the field the reference's KSFD.ksfdrandom.random_function produced from the same
    coarse samples (tests/golden/randfn_*.npz)"""
    z = load_golden(name)
    n = tuple(int(x) for x in z['n'])
    cfg = ProblemConfig.standard(len(n), n, L=tuple(float(x) for x in z['L']), nlig=1)
    k = klib.KSFDHip(cfg)
    k.set_state_random(z['z'] - 9000.0, rho0=9000.0)            # golden samples already carry the mean
    rho = k.get_state()[:cfg.N].reshape(n, order='F')
    assert np.abs(rho - z['out']).max() <= 1e-13 * np.abs(z['out']).max()
    k.close()


def test_nine_ligands_through_the_generic_kernels_vs_oracle():
    """fourier_series() expansions (KSFD/ksfdligand.py:315-388) can produce more ligand fields than a model names: 9 ligands in
    3 groups (F = 10) run the generic kernels (the strip kernels are instantiated for <= 4) and must match the oracle; 13 are
    rejected with KSFD_EINVAL (KSFD_MAX_LIG = 12)"""
    from oracle import ko
    nl = 9
    rng = np.random.default_rng(21)
    cfg = ProblemConfig(dim=2, n=(24, 20), L=(0.07, 0.06), lig_group=[l % 3 for l in range(nl)], lig_w=0.5 + rng.random(nl),
                        lig_s=0.005 + 0.01 * rng.random(nl), lig_gamma=0.005 + 0.01 * rng.random(nl), lig_D=1e-6 * (1 + rng.random(nl)),
                        grp_alpha=[1500.0, 1200.0, 1800.0], grp_beta=[5.56e-4, -3e-4, 2e-4])
    N = cfg.N
    rho = 9000 + 900 * rng.standard_normal(N)
    u = np.concatenate([rho] + [rho * cfg.lig_s[l] / cfg.lig_gamma[l] * (1 + 0.05 * rng.standard_normal(N)) for l in range(nl)])
    v = rng.standard_normal(u.size)
    o = ko.Oracle(cfg)
    k = klib.KSFDHip(cfg)
    assert rel_l2(k.rhs(u), o.rhs(u)) < 1e-12
    assert rel_l2(k.jvp(v, u), o.jvp(u, v)) < 1e-12
    k.set_state(u)
    assert rel_l2(k.jvp(v), o.jvp(u, v)) < 1e-12                  # frozen-coefficient path
    un, err, wr, _ = o.rosw_step(u, 0.05, 0.01, 1e-6, solver='gmres', ksp_rtol=1e-13)
    t, hn, st, rc = k.step(0.0, 0.05, klib.default_step_opts(adapt=0, atol=0.01, rtol=1e-6, ksp_rtol=1e-12))
    assert rel_l2(k.get_state(), un) < 1e-10
    k.close()
    with pytest.raises((klib.KSFDError, ValueError)):
        nl = 13
        klib.KSFDHip(ProblemConfig(dim=2, n=(16, 16), L=(0.05, 0.05), lig_group=[0] * nl, lig_w=[1.0] * nl, lig_s=[0.01] * nl,
                                   lig_gamma=[0.01] * nl, lig_D=[1e-6] * nl, grp_alpha=[1500.0], grp_beta=[5.56e-4]))
