"""CPU: pin the oracle (oracle/ksfd_oracle.c) against the golden vectors generated from the
reference's own symbolic + generated-C operator layer (tests/golden/make_golden.py)."""
import numpy as np
import pytest

from conftest import golden_cases, load_golden, rel_l2
from ksfd_amd.config import ProblemConfig
from ksfd_amd.layout import cijk_to_soa
from oracle import ko

OP_TOL = 1e-13        # rel-L2, SURVEY.md 8c


@pytest.mark.parametrize('name', golden_cases('op_'))
def test_oracle_operators(name):
    z = load_golden(name)
    cfg = ProblemConfig.from_golden(z)
    o = ko.Oracle(cfg)
    u = cijk_to_soa(z['u'])
    assert rel_l2(o.rhs(u), cijk_to_soa(z['rhs'])) < OP_TOL
    assert rel_l2(o.velocity(u), cijk_to_soa(z['vel'])) < OP_TOL
    assert rel_l2(o.jvp(u, cijk_to_soa(z['v'])), cijk_to_soa(z['Jv'])) < OP_TOL


def golden_entry_fields(z, cfg):
    """reference Jacobian entries as a dict (row_dof, di, dj, dk, col_dof) -> (nx,ny,nz) field (U rows: constants)"""
    S = tuple(int(x) for x in cfg.n[:cfg.dim])
    out = {}
    for off, val in zip(z['Jrho_off'], z['Jrho_val']):
        out[(0, int(off[0]), int(off[1]), int(off[2]), int(off[3]))] = np.asarray(val).reshape(S)
    for r, off, val in zip(z['JU_row'], z['JU_off'], z['JU_val']):
        out[(int(r), int(off[0]), int(off[1]), int(off[2]), int(off[3]))] = np.full(S, float(val))
    return out


def csr_entry_fields(cfg, rowptr, col, val):
    """the same dict from a CSR in the reference's Vec ordering (row = F*p + dof, p x-fastest)"""
    F, dim = cfg.nlig + 1, cfg.dim
    n = [int(x) for x in cfg.n[:3]]
    S = tuple(n[:dim])
    out = {}
    rows = np.repeat(np.arange(rowptr.size - 1), np.diff(rowptr))
    rp, rd = rows // F, rows % F
    cp, cd = col // F, col % F
    d = []
    for a, stride in enumerate((1, n[0], n[0] * n[1])):
        ra, ca = (rp // stride) % n[a], (cp // stride) % n[a]
        da = (ca - ra + n[a] // 2) % n[a] - n[a] // 2            # periodic offset in (-n/2, n/2]
        d.append(da)
    keys = np.stack([rd, d[0], d[1], d[2], cd], axis=1)
    for key in np.unique(keys, axis=0):
        sel = np.all(keys == key, axis=1)
        f = np.zeros(int(np.prod(S)))
        np.add.at(f, rp[sel], val[sel])
        out[tuple(int(x) for x in key)] = f.reshape(S, order='F')
    return out


@pytest.mark.parametrize('name', golden_cases('op_'))
def test_oracle_assembled_jacobian_entries(name):
    """every entry field of the reference's assembled Jacobian (rho rows: KSFD/ksfdsym.py:675-761,1067-1127; U rows
    :630-673) against the oracle's CSR; and CSR @ v == the matrix-free action"""
    import scipy.sparse as sp
    z = load_golden(name)
    cfg = ProblemConfig.from_golden(z)
    o = ko.Oracle(cfg)
    u = cijk_to_soa(z['u'])
    rowptr, col, val = o.jacobian_csr(u)
    ref = golden_entry_fields(z, cfg)
    got = csr_entry_fields(cfg, rowptr, col, val)
    scale = max(np.abs(f).max() for f in ref.values())
    for key, f in ref.items():
        if key not in got:
            assert np.abs(f).max() <= 1e-13 * scale, key          # the reference lists structurally zero entries too
            continue
        assert np.abs(got[key] - f).max() <= 1e-12 * max(np.abs(f).max(), 1e-300) + 1e-15 * scale, key
    for key, f in got.items():
        assert key in ref or np.abs(f).max() <= 1e-13 * scale, key
    F, N = cfg.nlig + 1, o.N
    A = sp.csr_matrix((val, col, rowptr), shape=(F * N, F * N))
    v = cijk_to_soa(z['v']).reshape(F, N)
    Jv = (A @ v.T.reshape(-1)).reshape(N, F).T.reshape(-1)       # PETSc (dof fastest) <-> SoA
    assert rel_l2(Jv, cijk_to_soa(z['Jv'])) < 1e-12


@pytest.mark.parametrize('name', golden_cases('op_'))
def test_oracle_groom_active(name):
    """inputs with negatives, sub-floor values and NaNs (KSFD/ksfdsym.py:888-900)"""
    z = load_golden(name)
    o = ko.Oracle(ProblemConfig.from_golden(z))
    ug = cijk_to_soa(z['ug'])
    assert np.isnan(ug).any() and (ug < 0).any()
    r = o.rhs(ug)
    assert np.isfinite(r).all()
    assert rel_l2(r, cijk_to_soa(z['rhs_g'])) < OP_TOL
    assert rel_l2(o.velocity(ug), cijk_to_soa(z['vel_g'])) < OP_TOL


STEP_TOL = 1e-11


@pytest.mark.parametrize('name', [n for n in golden_cases('step_') if 'manufactured' not in n and 'tdep' not in n])
def test_oracle_step_vs_reference_lu_golden(name):
    """oracle RA34PW2 (dense LU and GMRES) vs golden = reference operators + sparse LU + same tableau"""
    z = load_golden(name)
    cfg = ProblemConfig.from_golden(z)
    o = ko.Oracle(cfg)
    u = cijk_to_soa(z['u0'])
    h, atol, rtol = float(z['h']), float(z['atol']), float(z['rtol'])
    un, err, wr, _ = o.rosw_step(u, h, atol, rtol, solver='lu')
    assert rel_l2(un, cijk_to_soa(z['u1'])) < STEP_TOL
    assert rel_l2(err, cijk_to_soa(z['err1'])) < 1e-7
    assert abs(wr - z['wrms'][0]) <= 1e-7 * z['wrms'][0]
    ug, errg, wrg, its = o.rosw_step(u, h, atol, rtol, solver='gmres', ksp_rtol=1e-13)
    assert rel_l2(ug, un) < STEP_TOL and its > 0
    for s in range(1, int(z['nsteps'])):
        un, err, wr, _ = o.rosw_step(un, h, atol, rtol, solver='lu')
        assert abs(wr - z['wrms'][s]) <= 1e-6 * z['wrms'][s]
    assert rel_l2(un, cijk_to_soa(z['uN'])) < STEP_TOL


def test_oracle_operators_with_stage_time_parameters_vs_reference_golden():
    """step_2d_n1_tdep.npz (s_1_1, beta_1 expressions of t): the oracle's RHS with the stage-time tables and its assembled
    Jacobian with the step-start table, driven by the restated tableau + exact sparse LU, reproduce the reference-operator
    steps -- i.e. the per-stage semantics of ps.values(t) (KSFD/ksfdsym.py:1303-1312, 1430-1439)"""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    z = load_golden('step_2d_n1_tdep')

    def cfg_at(k, i):
        return ProblemConfig(dim=int(z['dim']), n=tuple(int(x) for x in z['n']), L=tuple(float(x) for x in z['L']),
                             lig_group=z['lig_group'], lig_w=z['tdep_lig_w'][k, i], lig_s=z['tdep_lig_s'][k, i],
                             lig_gamma=z['tdep_lig_gamma'][k, i], lig_D=z['tdep_lig_D'][k, i],
                             grp_alpha=z['tdep_grp_alpha'][k, i], grp_beta=z['tdep_grp_beta'][k, i],
                             s2=float(z['tdep_s2'][k, i]), rhomax=float(z['tdep_rhomax'][k, i]), cushion=float(z['tdep_cushion'][k, i]),
                             maxscale=float(z['tdep_maxscale'][k, i]), rhomin=float(z['rhomin']), Umin=float(z['Umin']),
                             cap_kind=int(z['cap_kind']))
    At, Gi, bt, b2t, asum = ko.tableau()
    gam = 1.0 / Gi[0, 0]
    u = cijk_to_soa(z['u0'])
    h = float(z['h'])
    F, N = int(z['nlig']) + 1, u.size // (int(z['nlig']) + 1)
    to_vec = lambda a: a.reshape(F, N).T.reshape(-1)             # SoA -> the CSR's unknown order F*p + dof
    to_soa = lambda x: x.reshape(N, F).T.reshape(-1)
    for k in range(int(z['nsteps'])):
        rp, col, val = ko.Oracle(cfg_at(k, 0)).jacobian_csr(u)
        J = sp.csr_matrix((val, col, rp), shape=(F * N, F * N))
        lu = spla.splu((sp.identity(F * N, format='csc') / (gam * h) - J).tocsc())
        Y = []
        for i in range(4):
            Z = u + sum(At[i, j] * Y[j] for j in range(i))
            Zdot = sum((Gi[i, j] / h) * Y[j] for j in range(i)) if i else 0.0
            rhs = ko.Oracle(cfg_at(k, i + 1)).rhs(Z) - Zdot
            Y.append(to_soa(lu.solve(to_vec(rhs))))
        u = u + sum(bt[j] * Y[j] for j in range(4))
        if k == 0:
            assert rel_l2(u, cijk_to_soa(z['u1'])) < STEP_TOL
    assert rel_l2(u, cijk_to_soa(z['uN'])) < STEP_TOL


def test_oracle_manufactured_known_answer():
    z = load_golden('step_1d_manufactured')
    cfg = ProblemConfig.from_golden(z)
    o = ko.Oracle(cfg)
    u = cijk_to_soa(z['u0'])
    for s in range(int(z['nsteps'])):
        src = [sv if np.any(sv) else None for i in range(4) for sv in z['src_v'][4 * s + i]]
        u, err, wr, _ = o.rosw_step(u, float(z['h']), float(z['atol']), float(z['rtol']), solver='lu', src_stage=src)
    assert rel_l2(u, cijk_to_soa(z['uN'])) < STEP_TOL
    assert np.abs(u - cijk_to_soa(z['exactN'])).max() < 2e-6


def test_tableau_order_conditions():
    """the RA34PW2 coefficients satisfy the order-3 / embedded order-2 conditions"""
    At, Gi, bt, b2t, asum = ko.tableau()
    G = np.linalg.inv(Gi)
    A = At @ G
    b, b2 = bt @ G, b2t @ G
    gam = G[0, 0]
    beta = (A + G).sum(axis=1)
    alpha = A.sum(axis=1)
    assert abs(b.sum() - 1) < 1e-14 and abs(b2.sum() - 1) < 1e-14
    assert abs(b @ beta - 0.5) < 1e-14
    assert abs(b @ alpha ** 2 - 1 / 3) < 1e-14
    assert abs(b @ (A + G) @ beta - 1 / 6) < 1e-14
    assert abs(b2 @ beta - 0.5) < 1e-14
    assert np.allclose((A + G)[3], b)          # stiffly accurate
    assert np.allclose(asum, alpha) and abs(gam - 4.3586652150845900e-01) < 1e-16


@pytest.mark.parametrize('name', golden_cases('adapt_'))
def test_oracle_adaptive_sequence_vs_reference_lu_golden(name):
    """reject/accept decisions, step sizes and states of an adaptive run from dt0=1e-8 (TSAdaptBasic restated)"""
    z = load_golden(name)
    cfg = ProblemConfig.from_golden(z)
    o = ko.Oracle(cfg)
    u = cijk_to_soa(z['u0'])
    t, h = 0.0, float(z['dt0'])
    atol, rtol = float(z['atol']), float(z['rtol'])
    for k in range(int(z['nsteps'])):
        prev, nrej = True, 0
        while True:
            un, err, wr, _ = o.rosw_step(u, h, atol, rtol, solver='lu')
            hn, acc = ko.adapt_basic(h, wr, prev_accept=prev)
            prev = acc
            if acc:
                break
            nrej += 1
            h = hn
        assert nrej == z['rej'][k]
        assert abs(h - z['h_acc'][k]) <= 1e-9 * z['h_acc'][k]
        assert abs(wr - z['wrms'][k]) <= 1e-6 * z['wrms'][k] + 1e-12
        u, t, h = un, t + h, hn
    assert abs(t - z['t_acc'][-1]) <= 1e-9 * t
    assert rel_l2(u, cijk_to_soa(z['uN'])) < 1e-10


@pytest.mark.parametrize('name', golden_cases('randfn_'))
def test_start_value_interpolation_vs_reference_random_function(name):
    """oracle (ko_random_function) and the host generator (ksfd_amd.initial) against the reference's own
    KSFD.ksfdrandom.random_function, run by tests/golden/make_randfn_golden.py"""
    from ksfd_amd.initial import smoothstep_interpolate
    z = load_golden(name)
    n, nc = tuple(int(x) for x in z['n']), tuple(int(x) for x in z['nc'])
    cfg = ProblemConfig.standard(len(n), n, L=tuple(float(x) for x in z['L']), nlig=1)
    got = ko.Oracle(cfg).random_function(list(nc) + [1] * (3 - len(n)), z['z'].ravel(order='F')).reshape(n, order='F')
    scale = np.abs(z['out']).max()
    assert np.abs(got - z['out']).max() <= 1e-14 * scale
    assert np.abs(smoothstep_interpolate(z['z'], n) - z['out']).max() <= 1e-14 * scale
