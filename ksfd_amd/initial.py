"""Initial values of (rho, U_1..U_n): the reference's start_values (ksfdsolver2.py:580-639), restated
with a separable interpolation instead of the reference's KDTree loop (KSFD/ksfdrandom.py:108-220,
which is O(hours) at 4096^2 and uses np.product, removed in numpy 2).

  coarse grid  n_d // 4 per axis (ksfdsolver2.py:581-586)
  z            rng.normal(size=coarse shape) * srho0 + Nworms / width**dim   (:595-610)
  rng          default_rng(SeedSequence(seed).spawn(size)[rank])            (ksfdrandom.py:44-49)
  interpolation weight f(x) = 2x^3 - 3x^2 + 1 on |dx|/h_coarse < 1 per axis, periodic (ksfdrandom.py:116,194-214);
               f(x) + f(1-x) = 1, so every fine point is a convex combination of its 2^dim coarse neighbours
  rho = rho0 + noise ; U_gl = rho * s_gl / gamma_gl                          (ksfdsolver2.py:617-637)

Parity: pinned.  tests/golden/randfn_*.npz hold fields produced by the reference's own random_function (run in the
build container on a data-only stand-in for its Grid, tests/golden/make_randfn_golden.py); this module, the oracle's
ko_random_function and the device kernel (ksfd_set_state_random) reproduce them to 1e-14.  The reference flattens its
coordinate arrays in C order but its Vecs in x-fastest order (ksfdrandom.py:183,193): on grids with the same point count
on every axis -- all its shipped runs -- the two reversals cancel and the result is the tensor-product interpolation
restated here; on other grids the reference returns a scramble of it, which is not reproduced.
"""
import numpy as np

DEFAULT_SEED = 793817931          # ksfdsolver2.py:412


def reference_rng(seed=DEFAULT_SEED, rank=0, size=1):
    return np.random.default_rng(np.random.SeedSequence(seed).spawn(size)[rank])


def _axis_weights(n, nc):
    xc = np.arange(n) * (nc / n)
    lo = np.floor(xc).astype(np.int64)
    fr = xc - lo
    f = lambda x: 2 * x ** 3 - 3 * x ** 2 + 1
    wlo = f(fr)
    whi = np.where(fr == 0.0, 0.0, f(1.0 - fr))
    return lo % nc, (lo + 1) % nc, wlo, whi


def smoothstep_interpolate(z, shape):
    """z: coarse samples indexed [i,j,k] (x first); returns fine array of `shape` indexed [i,j,k]."""
    out = np.asarray(z, dtype=np.float64)
    for ax, n in enumerate(shape):
        lo, hi, wlo, whi = _axis_weights(n, out.shape[ax])
        sh = [1] * out.ndim
        sh[ax] = n
        out = np.take(out, lo, axis=ax) * wlo.reshape(sh) + np.take(out, hi, axis=ax) * whi.reshape(sh)
    return out


def start_values(cfg, seed=DEFAULT_SEED, rho0=9000.0, srho0=90.0, Nworms=0.0, coarse=None, rng=None):
    """Returns the state as flat SoA (x fastest) of length F*N on the GLOBAL grid."""
    dim = cfg.dim
    shape = tuple(cfg.n[:dim])
    coarse = tuple(coarse) if coarse is not None else tuple(max(1, s // 4) for s in shape)
    rng = rng if rng is not None else reference_rng(seed)
    if srho0 == 0.0:
        noise = np.full(shape, Nworms / (cfg.L[0] ** dim))
    else:
        z = rng.normal(size=coarse) * srho0 + Nworms / (cfg.L[0] ** dim)
        noise = smoothstep_interpolate(z, shape)
    rho = rho0 + noise
    planes = [rho.ravel(order='F')]
    for l in range(cfg.nlig):
        planes.append(planes[0] * (cfg.lig_s[l] / cfg.lig_gamma[l]))
    return np.concatenate(planes)
