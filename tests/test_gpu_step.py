"""GPU: the implicit step (ROSW RA34PW2 + matrix-free GMRES) vs
  (a) golden states = REFERENCE operators + exact sparse LU (tests/golden/make_golden.py), and
  (b) the oracle's dense-LU / GMRES steps on seeded inputs.
PETSc itself is unavailable: against PETSc the stepper parity is unpinned (see oracle/ksfd_oracle.c)."""
import numpy as np
import pytest

from conftest import golden_cases, load_golden, rel_l2
from ksfd_amd.config import ProblemConfig
from ksfd_amd.layout import cijk_to_soa
from ksfd_amd import lib as klib
from oracle import ko

pytestmark = pytest.mark.gpu
STEP_TOL = 1e-10      # rel-L2 of the state; north_star asks 1e-8


def fixed_opts(z, **kw):
    return klib.default_step_opts(adapt=0, atol=float(z['atol']), rtol=float(z['rtol']), ksp_rtol=1e-12,
                                  ksp_max_it=4000, **kw)


@pytest.mark.parametrize('orth,tuning', [(0, 1), (1, 1), (0, 3), (0, 0), (0, 1 | 8), (0, 1 | 16), (0, 1 | 8192)])
@pytest.mark.parametrize('name', [n for n in golden_cases('step_') if 'manufactured' not in n and 'tdep' not in n])
def test_fixed_steps_vs_reference_lu_golden(name, orth, tuning):
    """orth 0: CGS2 with algebraic second projection, 1: classic CGS2; tuning bit0 fused kernels, bit1 recompute J,
    bit3 pipelined (device-resident) GMRES forced, bit4 no Krylov recycling, bit13 3-D RHS through the generic stencil pass
    instead of the z-marching strip kernel (the default)"""
    z = load_golden(name)
    cfg = ProblemConfig.from_golden(z)
    k = klib.KSFDHip(cfg)
    k.set_tuning(use_fused=tuning)
    k.set_state(cijk_to_soa(z['u0']))
    t, h = float(z['t0']), float(z['h'])
    opts = fixed_opts(z, reserved=orth)
    for s in range(int(z['nsteps'])):
        t, hn, st, rc = k.step(t, h, opts)
        assert st.accepted and hn == h
        assert abs(st.wrms - z['wrms'][s]) <= 1e-6 * z['wrms'][s] + 1e-12
        if s == 0:
            assert rel_l2(k.get_state(), cijk_to_soa(z['u1'])) < STEP_TOL
            e = k.last_error_vector()
            assert rel_l2(e, cijk_to_soa(z['err1'])) < 1e-6
    assert rel_l2(k.get_state(), cijk_to_soa(z['uN'])) < STEP_TOL
    assert abs(t - (float(z['t0']) + int(z['nsteps']) * h)) < 1e-12
    k.close()


def test_manufactured_solution_known_answer():
    """options93nx128dt1: the --source term makes rho = murho + arho e^{lamda t} sin(...) exact."""
    z = load_golden('step_1d_manufactured')
    cfg = ProblemConfig.from_golden(z)
    k = klib.KSFDHip(cfg)
    k.set_state(cijk_to_soa(z['u0']))
    t, h = 0.0, float(z['h'])
    opts = fixed_opts(z)
    n = int(z['nsteps'])
    for s in range(n):
        for i in range(4):                     # source fields at the four stage times of this step
            sv = z['src_v'][4 * s + i]
            for c in range(cfg.F):
                k.set_source(c, sv[c] if np.any(sv[c]) else None, stage=i)
        t, hn, st, rc = k.step(t, h, opts)
    u = k.get_state()
    assert rel_l2(u, cijk_to_soa(z['uN'])) < STEP_TOL
    # known answer: deviation from the exact manufactured solution stays at truncation level
    assert np.abs(u - cijk_to_soa(z['exactN'])).max() < 2e-6
    k.close()


@pytest.mark.parametrize('shape,nlig,L,h', [((48, 40), 1, (0.1, 0.1), 0.05), ((32, 32), 2, (0.2, 0.2), 0.1),
                                             ((24,), 1, (0.05,), 0.2)])
def test_step_vs_oracle_lu(shape, nlig, L, h):
    dim = len(shape)
    cfg = ProblemConfig.standard(dim, shape, L=L, nlig=nlig)
    rng = np.random.default_rng(11)
    N = int(np.prod(shape))
    rho = 9000 + 90 * rng.standard_normal(N)
    u = np.concatenate([rho] + [rho * cfg.lig_s[l] / cfg.lig_gamma[l] for l in range(nlig)])
    o = ko.Oracle(cfg)
    un, err, wr, _ = o.rosw_step(u, h, 0.01, 1e-6, solver='lu')
    k = klib.KSFDHip(cfg)
    k.set_state(u)
    t, hn, st, rc = k.step(0.0, h, klib.default_step_opts(adapt=0, atol=0.01, rtol=1e-6, ksp_rtol=1e-12))
    assert rel_l2(k.get_state(), un) < STEP_TOL
    assert abs(st.wrms - wr) <= 1e-6 * wr
    assert st.rhs_evals == 4 and st.linear_its > 0
    k.close()


def test_adaptive_controller_matches_oracle_formula():
    """TSAdaptBasic restated: accept iff wrms<=1, h_next = h*clip(0.9*wrms^(-1/3), 0.1, 5)."""
    cfg = ProblemConfig.standard(2, (32, 32), L=(0.2, 0.2))
    rng = np.random.default_rng(2)
    N = 32 * 32
    rho = 9000 + 90 * rng.standard_normal(N)
    u = np.concatenate([rho, rho])
    k = klib.KSFDHip(cfg)
    k.set_state(u)
    opts = klib.default_step_opts(adapt=1, atol=0.01, rtol=1e-6, ksp_rtol=1e-10)
    t, h = 0.0, 1e-8
    for s in range(12):
        t0 = t
        t, hn, st, rc = k.step(t, h, opts)
        assert st.accepted
        want, acc = ko.adapt_basic(st.h_used, st.wrms)
        assert acc and abs(hn - want) <= 1e-12 * want
        assert abs(t - (t0 + st.h_used)) < 1e-15 * max(1, abs(t))
        h = hn
    assert h > 1e-3       # the controller ramps up from dt0=1e-8 the way options84 runs do
    # force a rejection: absurdly large trial step with a tight tolerance
    tight = klib.default_step_opts(adapt=1, atol=1e-9, rtol=1e-12, ksp_rtol=1e-10, reserved=2)     # bit 1: single attempt per call
    before = k.get_state()
    t2, hn2, st2, rc2 = k.step(t, 50.0, tight)
    assert not st2.accepted and st2.rejections == 1 and hn2 < 50.0 and t2 == t
    assert np.array_equal(k.get_state(), ko.Oracle(cfg).groom(before))    # rolled back
    k.close()


@pytest.mark.parametrize('name', ['step_2d_n1_stiff', 'step_2d_n1_mild'])
def test_multigrid_preconditioned_steps_vs_reference_lu_golden(name):
    """pc_type=1: right-preconditioned GMRES with the geometric-multigrid V cycle gives the same steps"""
    z = load_golden(name)
    cfg = ProblemConfig.from_golden(z)
    k = klib.KSFDHip(cfg)
    k.set_state(cijk_to_soa(z['u0']))
    t, h = float(z['t0']), float(z['h'])
    opts = fixed_opts(z, pc_type=1)
    its = 0
    for s in range(int(z['nsteps'])):
        t, hn, st, rc = k.step(t, h, opts)
        its += st.linear_its
        if s == 0:
            assert rel_l2(k.get_state(), cijk_to_soa(z['u1'])) < STEP_TOL
    assert rel_l2(k.get_state(), cijk_to_soa(z['uN'])) < STEP_TOL
    k.close()


def test_multigrid_makes_very_stiff_steps_cheap():
    """h*gamma*lambda_max ~ 1e5: what the reference leaves to LU.  Multigrid keeps GMRES at a handful of iterations
    per stage; the result equals the oracle's dense-LU step."""
    cfg = ProblemConfig.standard(2, (32, 32), L=(0.08, 0.08), nlig=2)
    rng = np.random.default_rng(4)
    N = 32 * 32
    rho = 9000 + 90 * rng.standard_normal(N)       # srho0=90 as in the reference's runs (options84:31)
    u = np.concatenate([rho] + [rho * cfg.lig_s[l] / cfg.lig_gamma[l] for l in range(2)])
    h = 50.0
    un, err, wr, _ = ko.Oracle(cfg).rosw_step(u, h, 0.01, 1e-6, solver='lu')
    k = klib.KSFDHip(cfg)
    k.set_state(u)
    t, hn, st, rc = k.step(0.0, h, klib.default_step_opts(adapt=0, atol=0.01, rtol=1e-6, ksp_rtol=1e-11, pc_type=2))
    assert st.linear_its <= 4 * 16, st.linear_its
    assert rel_l2(k.get_state(), un) < 1e-9
    k.set_state(u)
    t, hn, st0, rc = k.step(0.0, h, klib.default_step_opts(adapt=0, atol=0.01, rtol=1e-6, ksp_rtol=1e-11, pc_type=0,
                                                           ksp_max_it=4000), raise_on_error=False)
    assert rc != 0 or st0.linear_its > 5 * st.linear_its        # unpreconditioned: far more work or no convergence
    k.close()


def test_long_adaptive_run_reaches_large_steps():
    """dt0=1e-8 -> the controller grows h by orders of magnitude (options80/81-style runs); auto pc keeps it solvable"""
    cfg = ProblemConfig.standard(2, (64, 64), L=(0.17, 0.17), nlig=1)
    rng = np.random.default_rng(9)
    N = 64 * 64
    rho = 9000 + 90 * rng.standard_normal(N)
    k = klib.KSFDHip(cfg)
    k.set_state(np.concatenate([rho, rho]))
    opts = klib.default_step_opts(adapt=1, atol=0.01, rtol=1e-6)
    t, h, its, rej = 0.0, 1e-8, [], 0
    for s in range(60):
        t, h, st, rc = k.step(t, h, opts)
        its.append(st.linear_its)
        rej += st.rejections
    assert t > 5.0 and h > 0.5, (t, h)
    assert max(its) < 400, its
    u = k.get_state()
    assert np.isfinite(u).all() and u.min() > 0
    k.close()


@pytest.mark.parametrize('name', golden_cases('adapt_'))
def test_adaptive_sequence_vs_reference_lu_golden(name):
    """ksfd_step's own reject/accept loop reproduces the adaptive run of 'reference operators + exact LU +
    TSAdaptBasic restated': same rejections, same accepted step sizes, same final state."""
    z = load_golden(name)
    cfg = ProblemConfig.from_golden(z)
    k = klib.KSFDHip(cfg)
    k.set_state(cijk_to_soa(z['u0']))
    opts = klib.default_step_opts(adapt=1, atol=float(z['atol']), rtol=float(z['rtol']), ksp_rtol=1e-12, ksp_max_it=4000)
    t, h = 0.0, float(z['dt0'])
    for s in range(int(z['nsteps'])):
        t, h, st, rc = k.step(t, h, opts)
        assert st.accepted and st.rejections == z['rej'][s]
        assert abs(st.h_used - z['h_acc'][s]) <= 1e-7 * z['h_acc'][s]
        assert abs(t - z['t_acc'][s]) <= 1e-9 * max(t, 1e-30)
    assert abs(h - float(z['h_next'])) <= 1e-6 * h
    assert rel_l2(k.get_state(), cijk_to_soa(z['uN'])) < 1e-9
    k.close()


def test_multigrid_3d_stiff_step_vs_oracle_lu():
    """3-D hierarchy (27-point full weighting / trilinear, z-marching Jacobian action on every level)"""
    cfg = ProblemConfig.standard(3, (16, 16, 16), L=(0.04, 0.04, 0.04), nlig=1)
    rng = np.random.default_rng(6)
    N = 16 ** 3
    rho = 9000 + 90 * rng.standard_normal(N)
    u = np.concatenate([rho, rho])
    h = 20.0
    un, err, wr, its_o = ko.Oracle(cfg).rosw_step(u, h, 0.01, 1e-6, solver='gmres', ksp_rtol=1e-12, maxit=6000)
    k = klib.KSFDHip(cfg)
    k.set_state(u)
    t, hn, st, rc = k.step(0.0, h, klib.default_step_opts(adapt=0, atol=0.01, rtol=1e-6, ksp_rtol=1e-11, pc_type=1))
    assert rel_l2(k.get_state(), un) < 1e-9
    assert st.linear_its <= 4 * 20, st.linear_its
    assert st.linear_its < its_o / 3
    k.close()


def test_failed_linear_solve_falls_back():
    """(a) unpreconditioned GMRES out of iterations -> the same system is re-solved with the multigrid preconditioner;
    (b) without multigrid and with the controller on, the step is rejected and quartered instead of aborting the run"""
    cfg = ProblemConfig.standard(2, (32, 32), L=(0.08, 0.08), nlig=1)
    rng = np.random.default_rng(12)
    rho = 9000 + 90 * rng.standard_normal(32 * 32)
    u = np.concatenate([rho, rho])
    k = klib.KSFDHip(cfg)
    # (a) h small enough that the stiffness estimate does not switch multigrid on by itself, iteration cap too low
    k.set_state(u)
    o = klib.default_step_opts(adapt=0, atol=0.01, rtol=1e-6, ksp_rtol=1e-9, ksp_max_it=12, pc_type=2)
    t, hn, st, rc = k.step(0.0, 0.02, o, raise_on_error=False)
    assert rc == 0 and st.accepted, k.last_error()
    un, _, _, _ = ko.Oracle(cfg).rosw_step(u, 0.02, 0.01, 1e-6, solver='lu')
    assert rel_l2(k.get_state(), un) < 1e-9
    # (b) no preconditioner allowed, controller on: quartered until 6 iterations are enough
    k.set_state(u)
    o = klib.default_step_opts(adapt=1, atol=0.01, rtol=1e-6, ksp_rtol=1e-8, ksp_max_it=6, pc_type=0)
    t, hn, st, rc = k.step(0.0, 0.02, o, raise_on_error=False)
    assert rc == 0 and st.accepted and st.rejections >= 1 and st.h_used <= 0.02 / 4
    # controller off: the failure is reported (reference semantics: SNES failure ends the run)
    k.set_state(u)
    o = klib.default_step_opts(adapt=0, atol=0.01, rtol=1e-6, ksp_rtol=1e-8, ksp_max_it=6, pc_type=0)
    t, hn, st, rc = k.step(0.0, 0.02, o, raise_on_error=False)
    assert rc == klib.ELINEAR and not st.accepted and np.array_equal(k.get_state(), ko.Oracle(cfg).groom(u))
    k.close()


def test_krylov_recycling_across_stages_saves_iterations_same_answer():
    """the four stage systems of a step share the matrix: projecting each new right-hand side onto the kept leading
    Arnoldi vectors of earlier stages (gmres(), stage >= 0) removes outer iterations and changes the result only
    within the solver tolerance; with a tight tolerance both variants equal the oracle's step"""
    n = 64
    cfg = ProblemConfig.standard(2, (n, n), L=(n * 4.0 / 1536, n * 4.0 / 1536), nlig=1)
    rng = np.random.default_rng(17)
    rho = 9000 + 90 * rng.standard_normal(n * n)
    u = np.concatenate([rho, rho])
    h = 0.02                                                    # h*gamma*lambda_max ~ 4: polynomial-preconditioned regime
    un, _, _, _ = ko.Oracle(cfg).rosw_step(u, h, 0.01, 1e-6, solver='gmres', ksp_rtol=1e-13, maxit=2000)   # dense LU would take minutes here
    k = klib.KSFDHip(cfg)
    k.set_spectral_params(enable=0)                             # this test is about the Krylov path (64^2 would otherwise take the spectral solver)
    res = {}
    for name, tune in (('on', 1), ('off', 1 | 16), ('all', 1 | 32)):
        k.set_tuning(use_fused=tune)
        for tol in (1e-6, 1e-11):
            k.set_state(u)
            t, hn, st, rc = k.step(0.0, h, klib.default_step_opts(adapt=0, atol=0.01, rtol=1e-6, ksp_rtol=tol))
            res[name, tol] = (st.linear_its, k.get_state())
    assert res['on', 1e-6][0] < res['off', 1e-6][0], (res['on', 1e-6][0], res['off', 1e-6][0])
    assert res['on', 1e-11][0] < res['off', 1e-11][0]
    for name in ('on', 'off', 'all'):
        assert rel_l2(res[name, 1e-11][1], un) < 1e-10, name
        assert rel_l2(res[name, 1e-6][1], un) < 1e-7, name
    k.close()


def test_stage_guesses_in_the_multigrid_regime_save_iterations_same_answer():
    """x0 = least-squares combination of the earlier stage solutions of the step (ksfd_step, multigrid branch): the correction
    equation is solved to the tolerance of the ORIGINAL system on the true residual b - A x0, so the step equals the oracle's
    LU step with the guesses on and off; with them the V-cycle iteration count of stages 2-4 drops"""
    cfg = ProblemConfig.standard(2, (32, 32), L=(0.08, 0.08), nlig=2)
    rng = np.random.default_rng(11)
    N = 32 * 32
    rho = 9000 + 90 * rng.standard_normal(N)
    u = np.concatenate([rho] + [rho * cfg.lig_s[l] / cfg.lig_gamma[l] for l in range(2)])
    h = 20.0
    un, _, _, _ = ko.Oracle(cfg).rosw_step(u, h, 0.01, 1e-6, solver='lu')
    res = {}
    for name, tune in (('on', 1), ('off', 1 | 16384)):
        for tol in (1e-6, 1e-11):
            k = klib.KSFDHip(cfg)            # a handle each: the hierarchy's set-up remembers the previous one (warm power iteration)
            k.set_tuning(use_fused=tune)
            k.set_state(u)
            t, hn, st, rc = k.step(0.0, h, klib.default_step_opts(adapt=0, atol=0.01, rtol=1e-6, ksp_rtol=tol, pc_type=1))
            assert st.pc_used & 2
            res[name, tol] = (st.linear_its, k.get_state())
            k.close()
    assert res['on', 1e-6][0] < res['off', 1e-6][0], (res['on', 1e-6][0], res['off', 1e-6][0])
    for name in ('on', 'off'):
        assert rel_l2(res[name, 1e-11][1], un) < 1e-9, name
        assert rel_l2(res[name, 1e-6][1], un) < 5e-6, name    # a residual tolerance: the state follows it to a small factor


@pytest.mark.parametrize('shape,nlig', [((64, 64), 1), ((40, 24), 2), ((32, 32, 32), 1)])
def test_checkpoint_restore_replays_the_same_steps(shape, nlig):
    """ksfd_checkpoint (what bench.py walks its pinned window with): state, and what the solvers remember from step to step, come back
    exactly -- the steps after a restore are bitwise the steps after the save (spectral, polynomial and plain regimes on the way)"""
    dim = len(shape)
    cfg = ProblemConfig.standard(dim, shape, L=tuple(n * 4.0 / 1536 for n in shape), nlig=nlig)
    rng = np.random.default_rng(31)
    rho = 9000 + 90 * rng.standard_normal(cfg.N)
    u = np.concatenate([rho] + [rho * cfg.lig_s[l] / cfg.lig_gamma[l] for l in range(nlig)])
    k = klib.KSFDHip(cfg)
    k.set_state(u)
    opts = klib.default_step_opts(adapt=1, atol=0.01, rtol=1e-6)
    t, h = 0.0, 1e-3
    for _ in range(4):
        t, h, st, rc = k.step(t, h, opts)
    k.checkpoint()
    tc, hc = t, h

    def run():
        tt, hh, out = tc, hc, []
        for _ in range(5):
            tt, hh, st, rc = k.step(tt, hh, opts)
            out.append((tt, hh, st.linear_its, st.rejections, st.pc_used, k.get_state().copy()))
        return out
    first = run()
    k.restore()
    second = run()
    for a, b in zip(first, second):
        assert a[:5] == b[:5]
        assert np.array_equal(a[5], b[5])
    k.close()


def test_checkpoint_restore_replays_the_same_steps_in_the_multigrid_regime():
    """same, with the V cycle as preconditioner (pc_type = 1, h ~ 20): what a multigrid set-up keeps for the next one (the power-iteration
    vector of every level) is dropped at the save and at the restore, so both replays set the hierarchy up from the same cold start"""
    cfg = ProblemConfig.standard(2, (64, 64), L=(0.16, 0.16), nlig=2)
    rng = np.random.default_rng(37)
    rho = 9000 + 90 * rng.standard_normal(cfg.N)
    u = np.concatenate([rho] + [rho * cfg.lig_s[l] / cfg.lig_gamma[l] for l in range(2)])
    k = klib.KSFDHip(cfg)
    k.set_state(u)
    opts = klib.default_step_opts(adapt=1, atol=0.01, rtol=1e-6, pc_type=1)
    t, h = 0.0, 5.0
    for _ in range(3):
        t, h, st, rc = k.step(t, h, opts)
        assert st.pc_used & 2
    k.checkpoint()
    tc, hc = t, h

    def run():
        tt, hh, out = tc, hc, []
        for _ in range(4):
            tt, hh, st, rc = k.step(tt, hh, opts)
            out.append((tt, hh, st.linear_its, st.rejections, st.pc_used, k.get_state().copy()))
        return out
    first = run()
    k.restore()
    second = run()
    for a, b in zip(first, second):
        assert a[:5] == b[:5]
        assert np.array_equal(a[5], b[5])
    k.close()


@pytest.mark.parametrize('restart', [5, 9])
def test_short_restart_lengths_give_the_same_step(restart):
    """large grids run with a restart length sized to the free HBM (ksfd_create); restarts + recycling with few slots must
    not change the answer"""
    n = 48
    cfg = ProblemConfig.standard(2, (n, n), L=(n * 4.0 / 1536, n * 4.0 / 1536), nlig=1)
    rng = np.random.default_rng(23)
    rho = 9000 + 90 * rng.standard_normal(n * n)
    u = np.concatenate([rho, rho])
    h = 0.05
    un, _, _, _ = ko.Oracle(cfg).rosw_step(u, h, 0.01, 1e-6, solver='gmres', ksp_rtol=1e-13, maxit=4000)
    k = klib.KSFDHip(cfg)
    for pc in (0, 2):
        k.set_state(u)
        t, hn, st, rc = k.step(0.0, h, klib.default_step_opts(adapt=0, atol=0.01, rtol=1e-6, ksp_rtol=1e-11, ksp_restart=restart,
                                                              ksp_max_it=4000, pc_type=pc))
        assert rel_l2(k.get_state(), un) < 1e-9, (pc, st.linear_its)
    k.close()


@pytest.mark.parametrize('shape', [(64, 48), (33, 20), (16, 16, 16)])
def test_fused_multigrid_smoother_gives_the_same_iterates(shape):
    """modes 5/6 of the Jacobian-action kernels (smoother algebra in the epilogue) vs the separate smoother kernels: the V cycle is the
    same linear operator, so iteration counts and the step agree (2-D strip kernel, generic kernel for odd nx, 3-D)"""
    dim = len(shape)
    cfg = ProblemConfig.standard(dim, shape, L=tuple(0.0025 * n for n in shape), nlig=2 if dim == 2 else 1)
    rng = np.random.default_rng(31)
    rho = 9000 + 90 * rng.standard_normal(cfg.N)
    u = np.concatenate([rho] + [rho * cfg.lig_s[l] / cfg.lig_gamma[l] for l in range(cfg.nlig)])
    k = klib.KSFDHip(cfg)
    out = {}
    for name, tune in (('fused', 1 | 262144), ('separate', 1 | 4096 | 262144)):      # bit 18: every set-up from the same cold start
        k.set_tuning(use_fused=tune)
        k.set_state(u)
        t, hn, st, rc = k.step(0.0, 10.0, klib.default_step_opts(adapt=0, atol=0.01, rtol=1e-6, ksp_rtol=1e-10, pc_type=1))
        out[name] = (st.linear_its, k.get_state())
    assert out['fused'][0] == out['separate'][0]
    assert rel_l2(out['fused'][1], out['separate'][1]) < 1e-11
    k.close()


@pytest.mark.parametrize('shape,nlig', [((32, 32), 2), ((64, 48), 1), ((96, 64), 3), ((64, 64), 4), ((36, 36), 2), ((256, 32), 1)])
def test_multigrid_cycle_with_fp32_level_vectors(shape, nlig):
    """ksp_rtol >= 1e-7: the V cycle keeps its level vectors in fp32 (mg_vcycle32; arithmetic fp64, GMRES and its true-residual test fp64).
    Same step as with the fp64 cycle (KSFD_TUNE bit 19) to the solve tolerance, iteration counts within one per stage, and (32^2, where the
    sparse LU takes seconds) the oracle's LU step to 5e-6 -- one to four ligands, a grid whose second level is already the coarsest one (36^2: only level 0 runs in fp32), a slab-shaped one."""
    L = tuple(0.0025 * n for n in shape)
    if nlig <= 2:
        cfg = ProblemConfig.standard(2, shape, L=L, nlig=nlig)
    else:                                   # attractant + repellent of options84, then two more of the same kinds in their own groups
        cfg = ProblemConfig(dim=2, n=shape, L=L, lig_group=[0, 1, 0, 1][:nlig], lig_w=[1.0, 1.0, 0.5, 0.7][:nlig],
                            lig_s=[0.01, 0.001, 0.003, 0.02][:nlig], lig_gamma=[0.01, 0.001, 0.004, 0.01][:nlig],
                            lig_D=[1e-6, 1e-5, 3e-6, 2e-6][:nlig], grp_alpha=[1500.0, 1500.0], grp_beta=[5.56e-4, -5.56e-4])
    rng = np.random.default_rng(41)
    rho = 9000 + 90 * rng.standard_normal(cfg.N)
    u = np.concatenate([rho] + [rho * cfg.lig_s[l] / cfg.lig_gamma[l] for l in range(nlig)])
    h = 10.0
    out = {}
    for name, tune in (('fp32', 1), ('fp64', 1 | 524288)):
        k = klib.KSFDHip(cfg)
        k.set_tuning(use_fused=tune)
        k.set_state(u)
        t, hn, st, rc = k.step(0.0, h, klib.default_step_opts(adapt=0, atol=0.01, rtol=1e-6, ksp_rtol=1e-7, pc_type=1))
        assert st.pc_used & 2
        out[name] = (st.linear_its, k.get_state())
        k.close()
    assert abs(out['fp32'][0] - out['fp64'][0]) <= 4, (out['fp32'][0], out['fp64'][0])
    assert rel_l2(out['fp32'][1], out['fp64'][1]) < 2e-7
    if shape == (32, 32):
        un, _, _, _ = ko.Oracle(cfg).rosw_step(u, h, 0.01, 1e-6, solver='lu')
        assert rel_l2(out['fp32'][1], un) < 5e-6


@pytest.mark.parametrize('n,h', [(96, 5.0), (384, 50.0), (130, 2.0)])
def test_multigrid_1d_stiff_step_vs_oracle_lu(n, h):
    """1-D hierarchy (weights 1/4, 1/2, 1/4 / linear interpolation; four of the six option files the reference ships are 1-D)"""
    cfg = ProblemConfig.standard(1, (n,), L=(n / 384.0,), nlig=2)
    rng = np.random.default_rng(8)
    rho = 9000 + 90 * rng.standard_normal(n)
    u = np.concatenate([rho] + [rho * cfg.lig_s[l] / cfg.lig_gamma[l] for l in range(2)])
    un, err, wr, _ = ko.Oracle(cfg).rosw_step(u, h, 0.01, 1e-6, solver='lu')
    k = klib.KSFDHip(cfg)
    k.set_state(u)
    t, hn, st, rc = k.step(0.0, h, klib.default_step_opts(adapt=0, atol=0.01, rtol=1e-6, ksp_rtol=1e-11, pc_type=1))
    assert rel_l2(k.get_state(), un) < 1e-9
    assert st.linear_its <= 4 * 30, st.linear_its
    k.close()


def test_default_ksp_rtol_keeps_fields_within_1e8_in_the_bench_regime():
    """The reference solves every stage system exactly (LU, options84:58-60); here the default is GMRES to ksp_rtol = 1e-6.
    On the bench problem at reduced size (512^2, options84 spacing/physics, bench start values), adaptive steps from
    dt = 0.01 to model time T = 3 -- the controller takes h past 0.2, through the polynomial-preconditioned regime into the
    stiff one, i.e. the regime the driver's bench measures -- the default must stay within the north-star tolerance
    (1e-8 rel-L2) of a ksp_rtol = 1e-12 run AT THE SAME MODEL TIME (the last step is cut to land on T), with the same
    accept/reject sequence.  The step sizes themselves agree only to ~1e-4: the embedded error estimate is a small
    difference of the stage vectors, so it amplifies the solver tolerance, and h_next ~ wrms^(-1/3) inherits that."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import build_problem
    from ksfd_amd.initial import start_values
    cfg = build_problem(512, 1)
    u0 = start_values(cfg)
    k = klib.KSFDHip(cfg)
    T = 3.0

    def run(rt):
        k.set_state(u0)
        o = klib.default_step_opts(adapt=1, atol=0.01, rtol=1e-6)
        if rt is not None:
            o.ksp_rtol = rt
        t, h, hs, rej = 0.0, 0.01, [], []
        while t + h < T and len(hs) < 60:
            t, h, st, rc = k.step(t, h, o)
            hs.append(st.h_used)
            rej.append(st.rejections)
        o.adapt = 0
        t, _, st, rc = k.step(t, T - t, o)                    # land exactly on T
        assert abs(t - T) < 1e-12
        return k.get_state(), np.array(hs), rej

    ref, hs_ref, rej_ref = run(1e-12)
    got, hs_got, rej_got = run(None)                            # library default
    k.close()
    assert hs_ref.max() > 0.2, hs_ref                           # the run really reaches the regime in question
    assert rej_got == rej_ref
    drift = float(np.abs(hs_got / hs_ref - 1.0).max())
    print('default ksp_rtol vs 1e-12 over %d adaptive steps to T = %g: step-size drift %.2e, fields %.2e' % (len(hs_ref), T, drift, rel_l2(got, ref)))
    assert drift < 1e-3, (hs_got, hs_ref)
    assert rel_l2(got, ref) < 1e-8
