"""GPU: the spectral preconditioner (ksfd_amd/csrc/spectral.hip.h).

It is a preconditioner, so the parity statements are (a) the operator itself against a numpy restatement (fft2 + the
closed-form arrow-block inverse) to fp32 accuracy, and (b) whole implicit steps solved WITH it against the oracle's exact
sparse-LU step to the usual 1e-10 -- GMRES stops on the true fp64 residual whatever the preconditioner does."""
import numpy as np
import pytest

from conftest import rel_l2
from ksfd_amd import lib as klib
from ksfd_amd.config import ProblemConfig
from oracle import ko

pytestmark = pytest.mark.gpu
GAMMA = 4.3586652150845900e-01


def _sym_d2(n, h):
    th = 2 * np.pi * np.fft.fftfreq(n)
    return (-30 + 32 * np.cos(th) - 2 * np.cos(2 * th)) / (12 * h * h)


def _numpy_spectral(cfg, u, shift, v):
    """(shift I - J0)^-1 v with J0 from the grid means of rho*G_rho, rho*G_Ul (tophat cap), exact 4th-order symbol"""
    nx, ny = cfg.n[0], cfg.n[1]
    F, nl = cfg.F, cfg.nlig
    ug = np.maximum(u.reshape(F, ny, nx), np.array([cfg.rhomin] + [cfg.Umin] * nl)[:, None, None])
    rho = ug[0]
    ms = cfg.maxscale * cfg.s2
    th = np.tanh((rho - cfg.rhomax) / cfg.cushion)
    a_rr = np.mean(rho * (cfg.s2 / rho + ms * (1 - th * th) / cfg.cushion))
    a_rU = []
    for l in range(nl):
        g = cfg.lig_group[l]
        ssum = cfg.grp_alpha[g] + sum(cfg.lig_w[m] * ug[m + 1] for m in range(nl) if cfg.lig_group[m] == g)
        a_rU.append(np.mean(rho * (-cfg.grp_beta[g] * cfg.lig_w[l] / ssum)))
    L2 = _sym_d2(nx, cfg.L[0] / nx)[None, :] + _sym_d2(ny, cfg.L[1] / ny)[:, None]
    vh = np.fft.fft2(v.reshape(F, ny, nx))
    d = [shift + cfg.lig_gamma[l] - cfg.lig_D[l] * L2 for l in range(nl)]
    den = shift - a_rr * L2 - sum(a_rU[l] * L2 * cfg.lig_s[l] / d[l] for l in range(nl))
    z0 = (vh[0] + sum(a_rU[l] * L2 / d[l] * vh[l + 1] for l in range(nl))) / den
    zs = [z0] + [(vh[l + 1] + cfg.lig_s[l] * z0) / d[l] for l in range(nl)]
    return np.real(np.fft.ifft2(np.array(zs))).reshape(-1)


def _lu_step(cfg, u, h, atol=0.01, rtol=1e-6):
    """one RA34PW2 step with the oracle's operators and an exact SPARSE LU of shift*I - J (the oracle's own rosw_step(solver='lu')
    factorises densely: minutes at 64^2); returns (unew, wrms)"""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    o = ko.Oracle(cfg)
    At, Gi, bt, b2t, asum = ko.tableau()
    gam = 1.0 / Gi[0, 0]
    F, N = cfg.F, cfg.N
    to_vec = lambda a: a.reshape(F, N).T.reshape(-1)
    to_soa = lambda x: x.reshape(N, F).T.reshape(-1)
    ug = o.groom(u)
    rp, col, val = o.jacobian_csr(ug)
    J = sp.csr_matrix((val, col, rp), shape=(F * N, F * N))
    lu = spla.splu((sp.identity(F * N, format='csc') / (gam * h) - J).tocsc())
    Y = []
    for i in range(4):
        Z = ug + sum(At[i, j] * Y[j] for j in range(i))
        Zdot = sum((Gi[i, j] / h) * Y[j] for j in range(i)) if i else 0.0
        Y.append(to_soa(lu.solve(to_vec(o.rhs(Z) - Zdot))))
    unew = ug + sum(bt[j] * Y[j] for j in range(4))
    err = sum((b2t[j] - bt[j]) * Y[j] for j in range(4))
    return unew, ko.wrms(unew, err, atol, rtol)


def _three_ligands(shape, L):
    """two ligands sharing group 0 (weights) + a repellent in its own group: F = 4 -> two complex pairs"""
    return ProblemConfig(dim=2, n=shape, L=L, lig_group=[0, 0, 1], lig_w=[1.0, 0.5, 1.0], lig_s=[0.01, 0.02, 0.001],
                         lig_gamma=[0.01, 0.03, 0.001], lig_D=[1e-6, 3e-6, 1e-5], grp_alpha=[1500.0, 1500.0],
                         grp_beta=[5.56e-4, -5.56e-4])


def _many_ligands(dim, shape, L, nl):
    """nl ligands in three groups (fourier_series()-style expansions): exercises the larger symbol blocks"""
    rng = np.random.default_rng(100 + nl)
    return ProblemConfig(dim=dim, n=shape, L=L, lig_group=[l % 3 for l in range(nl)], lig_w=0.5 + rng.random(nl),
                         lig_s=0.005 + 0.01 * rng.random(nl), lig_gamma=0.005 + 0.01 * rng.random(nl), lig_D=1e-6 * (1 + rng.random(nl)),
                         grp_alpha=[1500.0, 1200.0, 1800.0], grp_beta=[5.56e-4, -3e-4, 2e-4])


def _state(cfg, seed, amp=90.0):
    rng = np.random.default_rng(seed)
    rho = 9000.0 + amp * rng.standard_normal(cfg.N)
    return np.concatenate([rho] + [rho * cfg.lig_s[l] / cfg.lig_gamma[l] * (1 + 0.01 * rng.standard_normal(cfg.N)) for l in range(cfg.nlig)])


@pytest.mark.parametrize('shape,nlig', [((64, 32), 1), ((32, 128), 2), ((256, 512), 1), ((64, 64), 3), ((8192, 32), 1), ((32, 4096), 1),
                                        ((64, 64), 5), ((32, 64), 12),
                                        # extents 3 * 2^k (radix-3 stage): the reference's own 2-D grids are 384^2 and 1536^2 (options81:17, options84:15)
                                        ((384, 96), 1), ((96, 1536), 2), ((384, 384), 1), ((48, 96), 3), ((384, 256), 2), ((64, 1536), 1),
                                        ((6144, 32), 1), ((32, 6144), 1), ((192, 768), 5)])
@pytest.mark.parametrize('h', [0.02, 5.0])
def test_spectral_operator_vs_numpy(shape, nlig, h):
    L = tuple(n * 4.0 / 1536 for n in shape)
    cfg = _three_ligands(shape, L) if nlig == 3 else _many_ligands(2, shape, L, nlig) if nlig > 3 else ProblemConfig.standard(2, shape, L=L, nlig=nlig)
    u = _state(cfg, 3)
    v = np.random.default_rng(4).standard_normal(u.size)
    shift = 1.0 / (GAMMA * h)
    k = klib.KSFDHip(cfg)
    k.set_state(u)
    got = k.spectral_apply(shift, v)
    k.close()
    want = _numpy_spectral(cfg, u, shift, v)
    assert rel_l2(got, want) < 2e-5              # fp32 FFTs and symbol; it is a preconditioner


@pytest.mark.parametrize('shape,nlig,forced', [((32, 8192), 2, 0), ((64, 128), 2, 1), ((64, 64), 3, 1), ((32, 64), 5, 1),
                                               # one COLUMN per block (columns of more than 8192 points do not fit the LDS in pairs): 16384 rows for real,
                                               # forced on small grids with 1, 2 and 3 field pairs; (16384, 32): the rows of 16384 points (one row per block)
                                               ((32, 16384), 1, 0), ((16384, 32), 1, 0), ((64, 128), 1, 2), ((128, 64), 2, 2), ((64, 64), 3, 2), ((32, 64), 5, 2)])
def test_spectral_split_column_kernel_vs_numpy(shape, nlig, forced, monkeypatch):
    """more field pairs than the LDS holds columns for (8192 rows x 3 fields: 278 KB): the column pass runs as two launches over
    (column pair, field pair) blocks, the symbol stage reading the other field pairs' spectra from memory; KSFD_SPEC_SPLIT forces
    that path on small grids (= 2: split by column as well)"""
    if forced:
        monkeypatch.setenv('KSFD_SPEC_SPLIT', str(forced))
    L = tuple(n * 4.0 / 1536 for n in shape)
    cfg = _three_ligands(shape, L) if nlig == 3 else _many_ligands(2, shape, L, nlig) if nlig > 3 else ProblemConfig.standard(2, shape, L=L, nlig=nlig)
    u = _state(cfg, 3)
    v = np.random.default_rng(4).standard_normal(u.size)
    k = klib.KSFDHip(cfg)
    k.set_state(u)
    for h in (0.02, 5.0):
        shift = 1.0 / (GAMMA * h)
        assert rel_l2(k.spectral_apply(shift, v), _numpy_spectral(cfg, u, shift, v)) < 2e-5
    # and through a whole step (defect correction on top of it) against the sparse-LU step
    t, hn, st, rc = k.step(0.0, 0.3, klib.default_step_opts(adapt=0, atol=0.01, rtol=1e-6, ksp_rtol=1e-11, pc_type=4))
    assert st.pc_used & 8
    if cfg.N <= 64 * 128:
        un, _ = _lu_step(cfg, u, 0.3, 0.01, 1e-6)
        assert rel_l2(k.get_state(), un) < 1e-9
    k.close()


def test_spectral_unavailable_is_reported():
    cfg = ProblemConfig.standard(2, (48, 40), L=(0.1, 0.1))             # not powers of two
    k = klib.KSFDHip(cfg)
    k.set_state(_state(cfg, 1))
    with pytest.raises(klib.KSFDError):
        k.spectral_apply(1.0, np.zeros(2 * cfg.N))
    # pc_type 4 on such a handle: nothing spectral to use, the step still works
    t, h, st, rc = k.step(0.0, 0.05, klib.default_step_opts(adapt=0, atol=0.01, rtol=1e-6, pc_type=4))
    assert st.accepted and not (st.pc_used & klib.PC_SPECTRAL)
    k.close()


@pytest.mark.parametrize('shape,nlig,h', [((64, 32), 1, 0.05), ((32, 64), 2, 0.5), ((64, 64), 1, 20.0), ((64, 32), 3, 0.3),
                                          ((96, 48), 1, 0.3), ((48, 96), 2, 2.0), ((96, 64), 1, 0.05)])      # 3 * 2^k extents
@pytest.mark.parametrize('pc', [4, 2])
def test_step_with_spectral_preconditioner_vs_oracle_lu(shape, nlig, h, pc):
    L = tuple(n * 4.0 / 1536 for n in shape)
    cfg = _three_ligands(shape, L) if nlig == 3 else ProblemConfig.standard(2, shape, L=L, nlig=nlig)
    u = _state(cfg, 7)
    un, wr = _lu_step(cfg, u, h)
    k = klib.KSFDHip(cfg)
    k.set_state(u)
    t, hn, st, rc = k.step(0.0, h, klib.default_step_opts(adapt=0, atol=0.01, rtol=1e-6, ksp_rtol=1e-12, pc_type=pc))
    assert st.pc_used & klib.PC_SPECTRAL
    assert st.linear_its <= 4 * 18, st.linear_its           # to ksp_rtol 1e-12 (~3-4 per stage at the default 1e-6): a near-uniform state needs few sweeps at any h
    assert rel_l2(k.get_state(), un) < 1e-10
    assert abs(st.wrms - wr) <= 1e-6 * wr
    k.close()


def test_automatic_choice_leaves_the_spectral_preconditioner_when_coefficients_vary():
    """aggregated state (rho varies by a factor ~30): the constant-coefficient inverse no longer converges in a few
    iterations; pc_type 2 must notice, fall back inside the step (same answer as the LU oracle) and stay away afterwards"""
    shape = (64, 64)
    cfg = ProblemConfig.standard(2, shape, L=tuple(n * 4.0 / 1536 for n in shape), nlig=1)
    rng = np.random.default_rng(5)
    x = np.arange(64)
    bump = np.exp(-((x[None, :] - 20.0) ** 2 + (x[:, None] - 40.0) ** 2) / 30.0) + np.exp(-((x[None, :] - 50.0) ** 2 + (x[:, None] - 12.0) ** 2) / 20.0)
    rho = (800.0 + 24000.0 * bump).reshape(-1) * (1 + 0.01 * rng.standard_normal(cfg.N))
    u = np.concatenate([rho, rho * 1.0])
    h = 2.0
    un, wr = _lu_step(cfg, u, h)
    k = klib.KSFDHip(cfg)
    k.set_state(u)
    opts = klib.default_step_opts(adapt=0, atol=0.01, rtol=1e-6, ksp_rtol=1e-11)
    t, hn, st, rc = k.step(0.0, h, opts)
    assert rel_l2(k.get_state(), un) < 1e-9
    first = st.pc_used
    k.set_state(u)
    t, hn, st2, rc = k.step(0.0, h, opts)
    assert rel_l2(k.get_state(), un) < 1e-9
    if first & klib.PC_SPECTRAL and st.linear_its > 48:
        assert not (st2.pc_used & klib.PC_SPECTRAL)          # it backed off
    k.close()


def _numpy_spectral3d(cfg, u, shift, v):
    nx, ny, nz = cfg.n
    F, nl = cfg.F, cfg.nlig
    ug = np.maximum(u.reshape(F, nz, ny, nx), np.array([cfg.rhomin] + [cfg.Umin] * nl)[:, None, None, None])
    rho = ug[0]
    ms = cfg.maxscale * cfg.s2
    th = np.tanh((rho - cfg.rhomax) / cfg.cushion)
    a_rr = np.mean(rho * (cfg.s2 / rho + ms * (1 - th * th) / cfg.cushion))
    a_rU = []
    for l in range(nl):
        g = cfg.lig_group[l]
        ssum = cfg.grp_alpha[g] + sum(cfg.lig_w[m] * ug[m + 1] for m in range(nl) if cfg.lig_group[m] == g)
        a_rU.append(np.mean(rho * (-cfg.grp_beta[g] * cfg.lig_w[l] / ssum)))
    L2 = (_sym_d2(nx, cfg.L[0] / nx)[None, None, :] + _sym_d2(ny, cfg.L[1] / ny)[None, :, None] + _sym_d2(nz, cfg.L[2] / nz)[:, None, None])
    vh = np.fft.fftn(v.reshape(F, nz, ny, nx), axes=(1, 2, 3))
    d = [shift + cfg.lig_gamma[l] - cfg.lig_D[l] * L2 for l in range(nl)]
    den = shift - a_rr * L2 - sum(a_rU[l] * L2 * cfg.lig_s[l] / d[l] for l in range(nl))
    z0 = (vh[0] + sum(a_rU[l] * L2 / d[l] * vh[l + 1] for l in range(nl))) / den
    zs = [z0] + [(vh[l + 1] + cfg.lig_s[l] * z0) / d[l] for l in range(nl)]
    return np.real(np.fft.ifftn(np.array(zs), axes=(1, 2, 3))).reshape(-1)


@pytest.mark.parametrize('shape,nlig', [((32, 32, 32), 1), ((64, 32, 128), 2), ((32, 64, 32), 1), ((32, 32, 32), 4)])
@pytest.mark.parametrize('h', [0.02, 5.0])
def test_spectral_operator_3d_vs_numpy(shape, nlig, h):
    L = tuple(n * 4.0 / 1536 for n in shape)
    cfg = _many_ligands(3, shape, L, nlig) if nlig > 2 else ProblemConfig.standard(3, shape, L=L, nlig=nlig)
    u = _state(cfg, 3)
    v = np.random.default_rng(4).standard_normal(u.size)
    shift = 1.0 / (GAMMA * h)
    k = klib.KSFDHip(cfg)
    k.set_state(u)
    got = k.spectral_apply(shift, v)
    k.close()
    assert rel_l2(got, _numpy_spectral3d(cfg, u, shift, v)) < 2e-5


@pytest.mark.parametrize('pc', [4, 2])
def test_step_3d_with_spectral_solver_vs_oracle(pc):
    shape = (32, 32, 32)
    cfg = ProblemConfig.standard(3, shape, L=tuple(n * 4.0 / 1536 for n in shape), nlig=1)
    u = _state(cfg, 9)
    h = 0.1
    un, err, wr, _ = ko.Oracle(cfg).rosw_step(u, h, 0.01, 1e-6, solver='gmres', ksp_rtol=1e-13, maxit=4000)
    k = klib.KSFDHip(cfg)
    k.set_state(u)
    t, hn, st, rc = k.step(0.0, h, klib.default_step_opts(adapt=0, atol=0.01, rtol=1e-6, ksp_rtol=1e-12, pc_type=pc))
    assert st.pc_used & klib.PC_SPECTRAL
    assert st.linear_its <= 4 * 18, st.linear_its
    assert rel_l2(k.get_state(), un) < 1e-10
    k.close()


def test_predicted_last_sweep_is_verified_by_the_library_itself():
    """The spectral defect correction applies the LAST sweep of a solve on the contraction measured on the earlier sweeps instead of
    evaluating one more residual (ksp_rtol >= 1e-8).  KSFD_SPEC_VERIFY=1 makes the library evaluate that residual anyway and fail the solve
    (KSFD_ELINEAR) if it is above the tolerance: an adaptive run of the bench problem at 1024^2 from dt0 = 1e-8 through the ramp into the
    regime the bench measures must get through it, with most stage solves of the later steps ending on a predicted sweep -- and give the
    fields of the same run with every solve verified (opts.reserved bit 3) to 1e-9."""
    import os
    import subprocess
    import sys
    code = '''
import sys, numpy as np
sys.path.insert(0, %r)
from bench import build_problem
from ksfd_amd import lib as klib
from ksfd_amd.initial import reference_rng
cfg = build_problem(1024, 1)
zc = reference_rng().normal(size=(256, 256)) * 90.0
out = []
for flag in (0, 8):
    k = klib.KSFDHip(cfg)
    k.set_state_random(zc, 9000.0)
    o = klib.default_step_opts(adapt=1, atol=0.01, rtol=1e-6, reserved=flag)
    t, h, pred, res = 0.0, 1e-8, 0, 0
    for s in range(34):
        t, h, st, rc = k.step(t, h, o)
        pred += st.predicted_final
        res += st.residual_evals
    out.append((k.get_state(), t, pred, res))
    k.close()
(a, ta, pa, ra), (b, tb, pb, rb) = out
print('RESULT', pa, ra, pb, rb, abs(ta - tb) / tb, float(np.linalg.norm(a - b) / np.linalg.norm(b)))
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, KSFD_SPEC_VERIFY='1')
    r = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    assert 'ABOVE TOLERANCE' not in r.stderr
    line = [l for l in r.stdout.splitlines() if l.startswith('RESULT')][0].split()
    pred, res, pred_v, res_v, dt, err = int(line[1]), int(line[2]), int(line[3]), int(line[4]), float(line[5]), float(line[6])
    assert pred >= 30 and pred_v == 0                      # predicted sweeps happened, and the flag switches them off
    assert res < res_v                                     # ... and they are what saves residual evaluations
    assert r.stderr.count('predicted final sweep') == pred
    assert dt < 1e-6 and err < 1e-9


@pytest.mark.parametrize('shape,nlig', [((128, 96), 1), ((64, 64), 2), ((96, 48), 1)])
def test_guess_inner_products_from_the_rhs_epilogue(shape, nlig):
    """The stage guesses need <b_i, b_j> of a new stage right-hand side with the (up to two) before it.  They come out of the RHS kernel's
    store epilogue (per-wave partials beside ||b_i||^2); KSFD_TUNE bit 21 computes them in a pass of their own as before: same sweeps, same
    accepted steps, step sizes and states equal to what the rounding of the sums does to the guesses (1e-9 / 1e-10: the solves stop at 1e-6),
    fewer launches."""
    cfg = ProblemConfig.standard(2, shape, L=tuple(n * 4.0 / 1536 for n in shape), nlig=nlig)
    rng = np.random.default_rng(7)
    rho = 9000 + 90 * rng.standard_normal(cfg.N)
    u = np.concatenate([rho] + [rho * cfg.lig_s[l] / cfg.lig_gamma[l] for l in range(nlig)])
    out = {}
    for name, tune in (('epilogue', 1), ('own pass', 1 | 2097152)):
        k = klib.KSFDHip(cfg)
        k.set_tuning(use_fused=tune)
        k.set_state(u)
        o = klib.default_step_opts(adapt=1, atol=0.01, rtol=1e-6, pc_type=4)
        t, h, its, launches = 0.0, 0.05, 0, 0
        for _ in range(4):
            t, h, st, rc = k.step(t, h, o)
            assert st.pc_used & 8
            its += st.linear_its
            launches += st.launches
        out[name] = (t, its, launches, k.get_state())
        k.close()
    a, b = out['epilogue'], out['own pass']
    assert a[1] == b[1] and abs(a[0] - b[0]) <= 1e-9 * abs(b[0])
    assert a[2] < b[2]
    assert rel_l2(a[3], b[3]) < 1e-10


@pytest.mark.parametrize('shape,nlig,env', [((32, 32, 64), 1, {'KSFD_SPEC_FUSE3': '0'}), ((64, 32, 32), 2, {'KSFD_SPEC_FUSE3': '1'}),
                                           ((64, 128), 2, {'KSFD_SPEC_SPLIT': '1', 'KSFD_SPEC_FUSE': '4'}),
                                           ((128, 64), 1, {'KSFD_SPEC_SPLIT': '2', 'KSFD_SPEC_FUSE': '4'})])
def test_unfused_edge_stages_give_the_same_operator(shape, nlig, env):
    """The fused edge stages of the 3-D y/z kernels (KSFD_SPEC_FUSE3) and of the split column kernel (KSFD_SPEC_FUSE bits 0/1) are a
    re-ordering of the same transforms: with them switched off (the knobs are read once per process, hence the child) the operator
    z = M^-1 v agrees with the default build-up to fp32 rounding."""
    import os
    import subprocess
    import sys
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = '''
import sys, numpy as np
sys.path.insert(0, %r)
from ksfd_amd import lib as klib
from ksfd_amd.config import ProblemConfig
shape, nlig, out = %r, %d, sys.argv[1]
cfg = ProblemConfig.standard(len(shape), shape, L=tuple(0.003 * n for n in shape), nlig=nlig)
rng = np.random.default_rng(5)
rho = 9000 + 90 * rng.standard_normal(cfg.N)
u = np.concatenate([rho] + [rho * cfg.lig_s[l] / cfg.lig_gamma[l] for l in range(nlig)])
k = klib.KSFDHip(cfg)
k.set_state(u)
k.step(0.0, 0.01, klib.default_step_opts(adapt=0, atol=0.01, rtol=1e-6, pc_type=4))      # frozen planes + means
v = rng.standard_normal(cfg.F * cfg.N)
np.save(out, k.spectral_apply(3.0, v))
k.close()
''' % (root, shape, nlig)
    res = []
    with tempfile.TemporaryDirectory() as d:
        for i, extra in enumerate(({k_: v_ for k_, v_ in env.items() if k_ == 'KSFD_SPEC_SPLIT'}, env)):
            out = os.path.join(d, 'z%d.npy' % i)
            r = subprocess.run([sys.executable, '-c', code, out], env=dict(os.environ, **extra), capture_output=True, text=True, timeout=200)
            assert r.returncode == 0, r.stderr[-1500:]
            res.append(np.load(out))
    assert rel_l2(res[1], res[0]) < 2e-6
