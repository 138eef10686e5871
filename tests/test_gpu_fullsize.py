"""GPU, BASELINE.json sizes: parity through size-independent properties (the oracle cannot run at these sizes in seconds).

  * periodic translation invariance  f(shift u) = shift f(u), J(shift u) shift v = shift (J(u) v)
  * linearity of the Jacobian action and its consistency with a centred difference of the RHS
  * known answer on a uniform state
  * periodic replication: a 512^2 state tiled 8x8 is a 4096^2 state with the same spacing; one implicit step of the big
    problem must be the tiling of the small problem's step (which the small-size tests tie to the oracle / golden vectors)
"""
import numpy as np
import pytest

from conftest import rel_l2
from ksfd_amd import lib as klib
from ksfd_amd.config import ProblemConfig

pytestmark = pytest.mark.gpu
SP = 4.0 / 1536                      # options84 grid spacing


def _cfg(shape, nlig):
    return ProblemConfig.standard(len(shape), shape, L=tuple(n * SP for n in shape), nlig=nlig)


def _state(cfg, seed):
    rng = np.random.default_rng(seed)
    rho = 9000.0 + 90.0 * rng.standard_normal(cfg.N)
    parts = [rho] + [rho * cfg.lig_s[l] / cfg.lig_gamma[l] * (1.0 + 0.01 * rng.standard_normal(cfg.N)) for l in range(cfg.nlig)]
    return np.concatenate(parts)


def _roll(x, cfg, shifts):
    F = x.size // cfg.N
    a = x.reshape((F,) + tuple(reversed(cfg.n[:cfg.dim])))                       # (F, [nz,] ny, nx)
    for ax, s in enumerate(shifts):
        a = np.roll(a, s, axis=a.ndim - 1 - ax)
    return a.reshape(-1)


@pytest.mark.parametrize('shape,nlig,shifts', [((4096, 4096), 1, (130, 1717)), ((8192, 8192), 2, (4097, 33)), ((256, 256, 256), 1, (3, 129, 70))])
def test_translation_invariance_and_linearity_at_baseline_sizes(shape, nlig, shifts):
    cfg = _cfg(shape, nlig)
    u = _state(cfg, 1)
    k = klib.KSFDHip(cfg)
    f = k.rhs(u)
    fs = k.rhs(_roll(u, cfg, shifts))
    assert rel_l2(fs, _roll(f, cfg, shifts)) < 1e-13
    # uniform state: every difference vanishes -> f_rho = 0, f_U = -gamma U + s rho
    c = np.concatenate([np.full(cfg.N, 9000.0)] + [np.full(cfg.N, 8000.0 + 100 * l) for l in range(nlig)])
    fc = k.rhs(c)
    assert np.abs(fc[:cfg.N]).max() < 1e-9 * 9000.0 * cfg.s2 / SP ** 2
    for l in range(nlig):
        want = -cfg.lig_gamma[l] * (8000.0 + 100 * l) + cfg.lig_s[l] * 9000.0
        assert np.allclose(fc[(l + 1) * cfg.N:(l + 2) * cfg.N], want, rtol=1e-9, atol=0)
    del fc, fs, f
    rng = np.random.default_rng(2)
    v1, v2 = rng.standard_normal(u.size), rng.standard_normal(u.size)
    k.set_state(u)
    j1, j2 = k.jvp(v1), k.jvp(v2)
    assert rel_l2(k.jvp(0.3 * v1 - 1.7 * v2), 0.3 * j1 - 1.7 * j2) < 1e-12
    k.set_state(_roll(u, cfg, shifts))
    assert rel_l2(k.jvp(_roll(v1, cfg, shifts)), _roll(j1, cfg, shifts)) < 1e-13
    # J v against a centred difference of the RHS kernel (two different kernels, same operator)
    eps = 1e-3
    fd = (k.rhs(_roll(u, cfg, shifts) + eps * v2) - k.rhs(_roll(u, cfg, shifts) - eps * v2)) / (2 * eps)
    assert rel_l2(fd, k.jvp(v2)) < 1e-6
    k.close()


def test_step_of_a_tiled_state_is_the_tiled_step():
    """4096^2 implicit step == 8x8 tiling of the 512^2 step (same spacing, same physics)"""
    small, big = _cfg((512, 512), 1), _cfg((4096, 4096), 1)
    us = _state(small, 3)
    tile = lambda x: np.tile(x.reshape(2, 512, 512), (1, 8, 8)).reshape(-1)
    opts = klib.default_step_opts(adapt=0, atol=0.01, rtol=1e-6, ksp_rtol=1e-11)
    ks = klib.KSFDHip(small)
    ks.set_state(us)
    t, h, st_s, rc = ks.step(0.0, 0.01, opts)
    want = tile(ks.get_state())
    wrms_small = st_s.wrms
    ks.close()
    kb = klib.KSFDHip(big)
    kb.set_state(tile(us))
    t, h, st_b, rc = kb.step(0.0, 0.01, opts)
    got = kb.get_state()
    kb.close()
    assert rel_l2(got, want) < 1e-10
    assert abs(st_b.wrms - wrms_small) <= 1e-6 * wrms_small              # the WRMS norm is a mean: tiling leaves it alone
    # the step did something: rho moved by far more than the agreement asked for above
    assert rel_l2(got[:big.N], tile(us)[:big.N]) > 1e-6


def test_pinned_window_of_the_bench_at_4096_with_the_default_ksp_rtol():
    """Parity in the regime AND at the size the driver's bench measures.  The bench problem at 512^2 (options84 spacing/physics, the
    synthetic start values of SURVEY.md 8d) is advanced adaptively from dt0 = 1e-8 to the start of the pinned window (t* >= 0.35,
    ksp_rtol = 1e-12), that state is tiled 8x8 to 4096^2 -- same spacing, so the tiling is an exact solution of the big problem --
    and the window's five adaptive steps (h = 0.11 ... 0.32) are taken
      * on the 512^2 tile with ksp_rtol = 1e-12 (which the small-size tests tie to the oracle's exact sparse-LU step), and
      * on 4096^2 with the library DEFAULTS (ksp_rtol = 1e-6, spectral defect correction with its predicted last sweep),
    followed by one fixed step that lands both runs on the same model time.  Same accept/reject sequence, step sizes within 1e-3,
    fields within the north-star tolerance 1e-8 rel-L2 (the reference solves these systems exactly: options84:58-60)."""
    from ksfd_amd.initial import reference_rng
    small, big = _cfg((512, 512), 1), _cfg((4096, 4096), 1)
    zc = reference_rng().normal(size=(128, 128)) * 90.0
    tight = klib.default_step_opts(adapt=1, atol=0.01, rtol=1e-6, ksp_rtol=1e-12)
    dflt = klib.default_step_opts(adapt=1, atol=0.01, rtol=1e-6)                 # options84:18-19, everything else as shipped
    assert dflt.ksp_rtol == 1e-6 and dflt.pc_type == 2
    ks = klib.KSFDHip(small)
    ks.set_state_random(zc, 9000.0)
    t, h = 0.0, 1e-8
    while t < 0.35:
        t, h, st, rc = ks.step(t, h, tight)
    t_star, h_star, u_star = t, h, ks.get_state()

    def window(k, opts, t_land=None):
        t, h, hs, acc = t_star, h_star, [], []
        for _ in range(5):
            t, h, st, rc = k.step(t, h, opts)
            hs.append(st.h_used)
            acc.append((st.accepted, st.rejections))
        if t_land is None:
            t_land = t + 0.5 * h
        fixed = klib.default_step_opts(adapt=0, atol=0.01, rtol=1e-6, ksp_rtol=opts.ksp_rtol)
        t2, _, st, rc = k.step(t, t_land - t, fixed)
        assert abs(t2 - t_land) < 1e-12
        return np.array(hs), acc, t_land, st

    hs_ref, acc_ref, t_land, _ = window(ks, tight)
    want = _tile(ks.get_state(), 2, (512, 512), 8)
    ks.close()
    assert 0.1 < hs_ref.min() and hs_ref.max() > 0.25, hs_ref                    # the bench's window
    kb = klib.KSFDHip(big)
    kb.set_state(_tile(u_star, 2, (512, 512), 8))
    hs_got, acc_got, _, st_last = window(kb, dflt, t_land)
    got = kb.get_state()
    kb.close()
    assert st_last.pc_used & 8 and st_last.predicted_final > 0                   # the solver path the bench times
    drift = np.abs(hs_got / hs_ref - 1.0).max()
    err = rel_l2(got, want)
    print('pinned window at 4096^2, default ksp_rtol vs 512^2 tile at 1e-12: step-size drift %.2e, fields %.2e' % (drift, err))
    assert acc_got == acc_ref
    assert drift < 1e-3, (hs_got, hs_ref)
    assert err < 1e-8, err


def _tile(x, F, small_shape, reps):
    """periodic replication of a flat SoA state: (F, [nz,] ny, nx) tiled `reps` times along every axis"""
    a = x.reshape((F,) + tuple(reversed(small_shape)))
    return np.tile(a, (1,) + (reps,) * len(small_shape)).reshape(-1)


@pytest.mark.parametrize('small,reps,nlig,hs', [
    ((1024, 1024), 8, 2, (1e-3, 0.1)),      # BASELINE configs[3]: 8192^2, 3 fields; h = 1e-3 plain GMRES, h = 0.1 (X ~ 19) spectral solver, two-phase column kernel
    ((64, 64, 64), 8, 1, (1e-3, 0.05)),     # BASELINE configs[4]: 512^3, 2 fields, 13-point star; second step with the 3-D spectral solver
])
def test_full_size_configs_are_tilings_of_small_ones(small, reps, nlig, hs):
    """RHS, Jacobian action and whole implicit steps of the 8192^2 x 3 and 512^3 problems equal the periodic tiling of the
    1024^2 / 64^3 problem with the same spacing and physics (options84:32-46; 9/13-point star KSFD/ksfdsym.py:177-178),
    which the small-size tests tie to the golden vectors.  Host copies are freed as soon as they are compared."""
    big_shape = tuple(n * reps for n in small)
    cs, cb = _cfg(small, nlig), _cfg(big_shape, nlig)
    F = cs.F
    us = _state(cs, 5)
    vs = np.random.default_rng(6).standard_normal(us.size)
    opts = klib.default_step_opts(adapt=0, atol=0.01, rtol=1e-6, ksp_rtol=1e-11)
    ks = klib.KSFDHip(cs)
    want = {'rhs': ks.rhs(us), 'jvp': ks.jvp(vs, us)}
    ks.set_state(us)
    want['jvp_frozen'] = ks.jvp(vs)                       # the stepper's path: frozen coefficient planes of the resident state
    t = 0.0
    for i, h in enumerate(hs):
        t, _, st, rc = ks.step(t, h, opts)
        want['step%d' % i] = ks.get_state()
        want['wrms%d' % i] = st.wrms
    ks.close()
    kb = klib.KSFDHip(cb)
    ub = _tile(us, F, small, reps)
    got = kb.rhs(ub)
    assert rel_l2(got, _tile(want['rhs'], F, small, reps)) < 1e-13
    del got
    vb = _tile(vs, F, small, reps)
    got = kb.jvp(vb, ub)
    assert rel_l2(got, _tile(want['jvp'], F, small, reps)) < 1e-13
    del got
    kb.set_state(ub)
    del ub
    got = kb.jvp(vb)
    assert rel_l2(got, _tile(want['jvp_frozen'], F, small, reps)) < 1e-13
    del got, vb
    t = 0.0
    for i, h in enumerate(hs):
        t, _, st, rc = kb.step(t, h, opts)
        assert st.accepted
        if i == 1:
            assert st.pc_used & 8                 # the full-size problem took the spectral solver (pc_type 2 = automatic)
        assert abs(st.wrms - want['wrms%d' % i]) <= 1e-6 * want['wrms%d' % i]
        got = kb.get_state()
        assert rel_l2(got, _tile(want['step%d' % i], F, small, reps)) < 1e-10, (i, h)
        del got
    kb.close()
