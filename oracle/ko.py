"""ctypes wrapper around oracle/libksfd_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module
(the product package ksfd_amd never does).  Arrays are SoA, x fastest: shape (F, nz, ny, nx)
C-contiguous, or flat (F*N,).  See oracle/ksfd_oracle.c for reference citations.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class KoConfig(C.Structure):
    _fields_ = [('dim', C.c_int32), ('nlig', C.c_int32), ('ngroups', C.c_int32), ('cap_kind', C.c_int32),
                ('n', C.c_int64 * 3), ('L', C.c_double * 3),
                ('s2', C.c_double), ('rhomax', C.c_double), ('cushion', C.c_double),
                ('maxscale', C.c_double), ('rhomin', C.c_double), ('Umin', C.c_double),
                ('lig_group', C.POINTER(C.c_int32)),
                ('lig_w', C.POINTER(C.c_double)), ('lig_s', C.POINTER(C.c_double)),
                ('lig_gamma', C.POINTER(C.c_double)), ('lig_D', C.POINTER(C.c_double)),
                ('grp_alpha', C.POINTER(C.c_double)), ('grp_beta', C.POINTER(C.c_double))]


def build(force=False):
    so = os.path.join(_HERE, 'libksfd_oracle.so')
    src = os.path.join(_HERE, 'ksfd_oracle.c')
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', _HERE, '-s'])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        dp = C.POINTER(C.c_double)
        cp = C.POINTER(KoConfig)
        _LIB.ko_groom.argtypes = [cp, dp]
        _LIB.ko_G.argtypes = [cp, dp, dp]
        _LIB.ko_rhs.argtypes = [cp, dp, C.POINTER(dp), dp]
        _LIB.ko_jvp.argtypes = [cp, dp, dp, dp]
        _LIB.ko_velocity.argtypes = [cp, dp, dp]
        _LIB.ko_cfl.argtypes = [cp, dp, dp, dp]
        _LIB.ko_wrms.argtypes = [C.c_int64, dp, dp, C.c_double, C.c_double]
        _LIB.ko_wrms.restype = C.c_double
        _LIB.ko_adapt_basic.argtypes = [C.c_double, C.c_double, C.POINTER(C.c_int)] + [C.c_double] * 5 + [C.c_int, C.c_double]
        _LIB.ko_adapt_basic.restype = C.c_double
        _LIB.ko_tableau_get.argtypes = [dp] * 5
        _LIB.ko_rosw_step.argtypes = [cp, dp, C.c_double, C.POINTER(dp), C.c_double, C.c_double, C.c_int,
                                      C.c_double, C.c_double, C.c_int, C.c_int, dp, dp, dp, C.POINTER(C.c_int)]
        _LIB.ko_set_threads.argtypes = [C.c_int]
        _LIB.ko_random_function.argtypes = [cp, C.POINTER(C.c_int64), dp, dp]
    return _LIB


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class Oracle:
    """Holds a ko_config built from a ksfd_amd.config.ProblemConfig-like object (attributes
    dim, n, L, nlig, ngroups, cap_kind, s2, rhomax, cushion, maxscale, rhomin, Umin, lig_*, grp_*)."""

    def __init__(self, cfg):
        self.cfg = cfg
        self.F = int(cfg.nlig) + 1
        self.N = int(cfg.n[0]) * int(cfg.n[1]) * int(cfg.n[2])
        self._keep = [np.ascontiguousarray(cfg.lig_group, dtype=np.int32)] + [
            np.ascontiguousarray(getattr(cfg, k), dtype=np.float64)
            for k in ('lig_w', 'lig_s', 'lig_gamma', 'lig_D', 'grp_alpha', 'grp_beta')]
        k = KoConfig()
        k.dim, k.nlig, k.ngroups, k.cap_kind = int(cfg.dim), int(cfg.nlig), int(cfg.ngroups), int(cfg.cap_kind)
        for a in range(3):
            k.n[a] = int(cfg.n[a])
            k.L[a] = float(cfg.L[a])
        for nm in ('s2', 'rhomax', 'cushion', 'maxscale', 'rhomin', 'Umin'):
            setattr(k, nm, float(getattr(cfg, nm)))
        k.lig_group = self._keep[0].ctypes.data_as(C.POINTER(C.c_int32))
        for nm, arr in zip(('lig_w', 'lig_s', 'lig_gamma', 'lig_D', 'grp_alpha', 'grp_beta'), self._keep[1:]):
            setattr(k, nm, _dp(arr))
        self.k = k

    def _in(self, a, planes=None):
        a = np.ascontiguousarray(a, dtype=np.float64).reshape(-1)
        assert a.size == (planes or self.F) * self.N, (a.size, planes, self.F, self.N)
        return a

    def _src_ptrs(self, src, nstage=1):
        if src is None:
            return None, None
        keep, arr = [], (C.POINTER(C.c_double) * (nstage * self.F))()
        for i, s in enumerate(src):
            if s is None:
                arr[i] = None
            else:
                s = np.ascontiguousarray(s, dtype=np.float64).reshape(-1)
                assert s.size == self.N
                keep.append(s)
                arr[i] = _dp(s)
        return arr, keep

    def groom(self, u):
        u = self._in(u).copy()
        lib().ko_groom(C.byref(self.k), _dp(u))
        return u

    def rhs(self, u, src=None):
        u = self._in(u)
        out = np.empty_like(u)
        arr, keep = self._src_ptrs(src)
        rc = lib().ko_rhs(C.byref(self.k), _dp(u), arr, _dp(out))
        assert rc == 0
        return out

    def jvp(self, u, v):
        u, v = self._in(u), self._in(v)
        out = np.empty_like(u)
        assert lib().ko_jvp(C.byref(self.k), _dp(u), _dp(v), _dp(out)) == 0
        return out

    def jacobian_csr(self, u):
        """(rowptr, col, val) of df/du at groom(u); unknowns ordered F*p + dof like the reference's Vec"""
        u = self._in(u)
        L = lib()
        L.ko_jacobian_nnz.restype = C.c_int64
        nnz = int(L.ko_jacobian_nnz(C.byref(self.k)))
        rowptr = np.empty(self.F * self.N + 1, dtype=np.int64)
        col = np.empty(nnz, dtype=np.int64)
        val = np.empty(nnz)
        ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int64))
        assert L.ko_jacobian_csr(C.byref(self.k), _dp(u), ip(rowptr), ip(col), _dp(val)) == 0
        return rowptr, col, val

    def velocity(self, u):
        u = self._in(u)
        out = np.empty(int(self.cfg.dim) * self.N)
        assert lib().ko_velocity(C.byref(self.k), _dp(u), _dp(out)) == 0
        return out

    def cfl(self, u):
        u = self._in(u)
        vmax = np.zeros(3)
        h = C.c_double()
        assert lib().ko_cfl(C.byref(self.k), _dp(u), _dp(vmax), C.byref(h)) == 0
        return vmax, h.value

    def rosw_step(self, u, h, atol, rtol, solver='lu', src_stage=None, ksp_rtol=1e-10, ksp_atol=0.0,
                  restart=30, maxit=2000):
        """src_stage: list of 4*F entries ([stage][field]) or None.  Returns unew, err, wrms, its."""
        u = self._in(u)
        unew, err = np.empty_like(u), np.empty_like(u)
        wr, its = C.c_double(), C.c_int()
        arr, keep = self._src_ptrs(src_stage, nstage=4)
        rc = lib().ko_rosw_step(C.byref(self.k), _dp(u), float(h), arr, float(atol), float(rtol),
                                0 if solver == 'lu' else 1, float(ksp_rtol), float(ksp_atol), int(restart),
                                int(maxit), _dp(unew), _dp(err), C.byref(wr), C.byref(its))
        if rc:
            raise RuntimeError('ko_rosw_step failed rc=%d' % rc)
        return unew, err, wr.value, its.value

    def random_function(self, nc, z):
        z = np.ascontiguousarray(z, dtype=np.float64).reshape(-1)
        ncs = (C.c_int64 * 3)(*[int(x) for x in nc])
        out = np.empty(self.N)
        lib().ko_random_function(C.byref(self.k), ncs, _dp(z), _dp(out))
        return out


def set_threads(n):
    """OpenMP threads for the operator loops (cpu_baseline leg); default 1."""
    lib().ko_set_threads(int(n))


def wrms(unew, err, atol, rtol):
    unew = np.ascontiguousarray(unew, dtype=np.float64).reshape(-1)
    err = np.ascontiguousarray(err, dtype=np.float64).reshape(-1)
    return lib().ko_wrms(unew.size, _dp(unew), _dp(err), float(atol), float(rtol))


def adapt_basic(h, enorm, safety=0.9, clip=(0.1, 5.0), dt_min=1e-20, dt_max=1e4, prev_accept=True, reject_safety=0.5):
    acc = C.c_int()
    hn = lib().ko_adapt_basic(float(h), float(enorm), C.byref(acc), safety, clip[0], clip[1], dt_min, dt_max,
                              int(bool(prev_accept)), reject_safety)
    return hn, bool(acc.value)


def tableau():
    At, Gi = np.zeros((4, 4)), np.zeros((4, 4))
    bt, b2t, asum = np.zeros(4), np.zeros(4), np.zeros(4)
    lib().ko_tableau_get(_dp(At), _dp(Gi), _dp(bt), _dp(b2t), _dp(asum))
    return At, Gi, bt, b2t, asum
