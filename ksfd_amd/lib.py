"""ctypes binding of libksfd_hip.so (include/ksfd_hip.h).

This is the binding INTEGRATION.md shows for the reference side.  There is no CPU fallback: if the
shared library is missing or no MI355X is visible, construction raises.
"""
import ctypes as C
import os

import numpy as np

from .config import CConfig, ProblemConfig
from .layout import PETSC, SOA, HDF5  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libksfd_hip.so')
NKCLASS = 14

OK, EINVAL, EHIP, ENOMEM, ELINEAR, ENAN, EREJECT, ECOMM = range(8)

# kernel classes of ksfd_profile / ksfd_bench_kernel
KC_RHS, KC_JVP, KC_MULTIDOT, KC_GSUPDATE, KC_LINCOMB, KC_BASISAXPY, KC_FINISH, KC_REDUCE, \
    KC_GFIELD, KC_VELOCITY, KC_MISC, KC_HALO, KC_MG, KC_SPECTRAL = range(14)
PC_NONE, PC_MULTIGRID, PC_POLYNOMIAL, PC_SPECTRAL = 1, 2, 4, 8      # bits of StepStats.pc_used


class KSFDError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__('ksfd_hip error %d: %s' % (code, msg))
        self.code = code


EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double),
                          C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int64)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int32, C.c_int32)
ALLTOALL_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64)


class CDist(C.Structure):
    _fields_ = [('rank', C.c_int32), ('size', C.c_int32), ('transport', C.c_int32), ('device', C.c_int32),
                ('nccl_id', C.c_void_p), ('exchange', EXCHANGE_FN), ('allreduce', ALLREDUCE_FN),
                ('ctx', C.c_void_p), ('alltoall', ALLTOALL_FN)]


class StepOpts(C.Structure):
    _fields_ = [('rtol', C.c_double), ('atol', C.c_double), ('adapt', C.c_int32), ('max_reject', C.c_int32),
                ('clip_lo', C.c_double), ('clip_hi', C.c_double), ('dt_min', C.c_double), ('dt_max', C.c_double),
                ('safety', C.c_double), ('reject_safety', C.c_double),
                ('ksp_rtol', C.c_double), ('ksp_atol', C.c_double),
                ('ksp_restart', C.c_int32), ('ksp_max_it', C.c_int32), ('pc_type', C.c_int32),
                ('reserved', C.c_int32)]


class StepStats(C.Structure):
    _fields_ = [('accepted', C.c_int32), ('rejections', C.c_int32), ('linear_its', C.c_int32),
                ('rhs_evals', C.c_int32), ('jvp_evals', C.c_int32), ('pc_used', C.c_int32),
                ('wrms', C.c_double), ('h_used', C.c_double), ('ksp_resid', C.c_double), ('bytes', C.c_double),
                ('launches', C.c_int32), ('host_syncs', C.c_int32), ('residual_evals', C.c_int32),
                ('predicted_final', C.c_int32)]


class Profile(C.Structure):
    _fields_ = [('ms', C.c_double * NKCLASS), ('bytes', C.c_double * NKCLASS), ('launches', C.c_int64 * NKCLASS),
                ('alg_bytes', C.c_double * NKCLASS)]


_lib = None

# every symbol include/ksfd_hip.h declares (checked by tests/test_abi.py without a GPU)
ABI_SYMBOLS = [
    'ksfd_kernel_class_name', 'ksfd_rccl_unique_id', 'ksfd_create', 'ksfd_destroy', 'ksfd_last_error', 'ksfd_update_params', 'ksfd_set_stage_params',
    'ksfd_local_range', 'ksfd_local_size', 'ksfd_set_state', 'ksfd_get_state', 'ksfd_device_state',
    'ksfd_device_plane_stride', 'ksfd_device_interior_offset', 'ksfd_set_source', 'ksfd_rhs', 'ksfd_jvp',
    'ksfd_velocity', 'ksfd_velocity_max', 'ksfd_groom', 'ksfd_count_worms', 'ksfd_scale_rho', 'ksfd_mul_rho', 'ksfd_jacobian_nnz', 'ksfd_jacobian_csr', 'ksfd_set_state_random', 'ksfd_snapshot_begin', 'ksfd_snapshot_wait', 'ksfd_checkpoint',
    'ksfd_default_step_opts', 'ksfd_step', 'ksfd_get_last_error_vector', 'ksfd_set_profiling',
    'ksfd_get_profile', 'ksfd_synchronize', 'ksfd_bench_kernel', 'ksfd_set_tuning', 'ksfd_set_mg_params', 'ksfd_set_poly_params',
    'ksfd_spectral_apply', 'ksfd_set_spectral_params',
]


def load():
    """dlopen libksfd_hip.so and declare signatures.  Raises if the library was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise KSFDError(EHIP, 'libksfd_hip.so not built (run `python -c "import __graft_entry__ as g; g.build()"` '
                              'or `make -C ksfd_amd/csrc`); there is no CPU fallback')
    L = C.CDLL(LIB_PATH)
    dp, vp = C.POINTER(C.c_double), C.c_void_p
    L.ksfd_kernel_class_name.restype = C.c_char_p
    L.ksfd_kernel_class_name.argtypes = [C.c_int32]
    L.ksfd_rccl_unique_id.argtypes = [vp]
    L.ksfd_create.argtypes = [C.POINTER(CConfig), C.POINTER(CDist), C.POINTER(vp)]
    L.ksfd_destroy.argtypes = [vp]
    L.ksfd_destroy.restype = None
    L.ksfd_last_error.argtypes = [vp]
    L.ksfd_last_error.restype = C.c_char_p
    L.ksfd_update_params.argtypes = [vp, C.POINTER(CConfig)]
    L.ksfd_set_stage_params.argtypes = [vp, C.c_int32, C.POINTER(CConfig)]
    L.ksfd_local_range.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.ksfd_local_size.argtypes = [vp]
    L.ksfd_local_size.restype = C.c_int64
    L.ksfd_set_state.argtypes = [vp, dp, C.c_int32]
    L.ksfd_get_state.argtypes = [vp, dp, C.c_int32]
    L.ksfd_device_state.argtypes = [vp]
    L.ksfd_device_state.restype = vp
    L.ksfd_device_plane_stride.argtypes = [vp]
    L.ksfd_device_plane_stride.restype = C.c_int64
    L.ksfd_device_interior_offset.argtypes = [vp]
    L.ksfd_device_interior_offset.restype = C.c_int64
    L.ksfd_set_source.argtypes = [vp, C.c_int32, C.c_int32, dp, C.c_int32]
    L.ksfd_rhs.argtypes = [vp, C.c_double, dp, dp, C.c_int32]
    L.ksfd_jvp.argtypes = [vp, dp, dp, dp, C.c_int32]
    L.ksfd_velocity.argtypes = [vp, dp, dp, C.c_int32]
    L.ksfd_velocity_max.argtypes = [vp, dp]
    L.ksfd_groom.argtypes = [vp]
    L.ksfd_count_worms.argtypes = [vp, dp]
    L.ksfd_scale_rho.argtypes = [vp, C.c_double]
    L.ksfd_mul_rho.argtypes = [vp, dp]
    L.ksfd_snapshot_begin.argtypes = [vp, C.c_int32, C.POINTER(C.c_int32)]
    L.ksfd_snapshot_wait.argtypes = [vp, C.c_int32, C.POINTER(C.POINTER(C.c_double))]
    L.ksfd_checkpoint.argtypes = [vp, C.c_int32]
    L.ksfd_set_state_random.argtypes = [vp, C.POINTER(C.c_int64), dp, C.c_double]
    L.ksfd_jacobian_nnz.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.ksfd_jacobian_csr.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64), dp]
    L.ksfd_default_step_opts.argtypes = [C.POINTER(StepOpts)]
    L.ksfd_default_step_opts.restype = None
    L.ksfd_step.argtypes = [vp, dp, dp, C.POINTER(StepOpts), C.POINTER(StepStats)]
    L.ksfd_get_last_error_vector.argtypes = [vp, dp, C.c_int32]
    L.ksfd_set_profiling.argtypes = [vp, C.c_int32]
    L.ksfd_get_profile.argtypes = [vp, C.POINTER(Profile), C.c_int32]
    L.ksfd_synchronize.argtypes = [vp]
    L.ksfd_bench_kernel.argtypes = [vp, C.c_int32, C.c_int32, dp, dp]
    L.ksfd_set_tuning.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32]
    L.ksfd_set_mg_params.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_double]
    L.ksfd_set_poly_params.argtypes = [vp, C.c_int32, C.c_double, C.c_double]
    L.ksfd_spectral_apply.argtypes = [vp, C.c_double, dp, dp, C.c_int32]
    L.ksfd_set_spectral_params.argtypes = [vp, C.c_double, C.c_int32]
    _lib = L
    return L


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def rccl_unique_id():
    """128-byte ncclUniqueId (call on rank 0, broadcast to the other ranks)."""
    buf = C.create_string_buffer(128)
    rc = load().ksfd_rccl_unique_id(C.cast(buf, C.c_void_p))
    if rc:
        raise KSFDError(rc, (load().ksfd_last_error(None) or b'').decode())
    return bytes(buf.raw)


def default_step_opts(**kw):
    o = StepOpts()
    load().ksfd_default_step_opts(C.byref(o))
    for k, v in kw.items():
        if not hasattr(o, k):
            raise TypeError('unknown step option %r' % k)
        setattr(o, k, v)
    return o


class KSFDHip:
    """Owns one ksfd_handle (one GPU / one slab).  Host arrays are numpy float64, flat, local slab."""

    def __init__(self, cfg: ProblemConfig, dist: CDist = None):
        self.L = load()
        self.cfg = cfg
        self._ccfg = cfg.as_ctypes()
        self._dist = dist
        h = C.c_void_p()
        rc = self.L.ksfd_create(C.byref(self._ccfg), C.byref(dist) if dist is not None else None, C.byref(h))
        if rc:
            raise KSFDError(rc, (self.L.ksfd_last_error(None) or b'').decode())
        self.h = h
        self.F = cfg.F
        self.nlocal = int(self.L.ksfd_local_size(h))        # F * local points
        b, e = C.c_int64(), C.c_int64()
        self.L.ksfd_local_range(h, C.byref(b), C.byref(e))
        self.slow_range = (b.value, e.value)

    def close(self):
        if getattr(self, 'h', None):
            self.L.ksfd_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc:
            raise KSFDError(rc, (self.L.ksfd_last_error(self.h) or b'').decode())

    def _vec(self, a, n=None):
        a = np.ascontiguousarray(a, dtype=np.float64).reshape(-1)
        if a.size != (n if n is not None else self.nlocal):
            raise ValueError('expected %d doubles, got %d' % (n if n is not None else self.nlocal, a.size))
        return a

    # ---- state
    def set_state(self, u, layout=SOA):
        u = self._vec(u)
        self._chk(self.L.ksfd_set_state(self.h, _dp(u), layout))

    def get_state(self, layout=SOA):
        out = np.empty(self.nlocal)
        self._chk(self.L.ksfd_get_state(self.h, _dp(out), layout))
        return out

    def update_params(self, cfg: ProblemConfig):
        self.cfg = cfg
        self._ccfg = cfg.as_ctypes()
        self._chk(self.L.ksfd_update_params(self.h, C.byref(self._ccfg)))

    def set_stage_params(self, stage, cfg=None):
        """ps.values(t_stage) for the stage-th RHS evaluation of the next steps (None clears; stage -1 = all four)"""
        if cfg is None:
            self._chk(self.L.ksfd_set_stage_params(self.h, int(stage), None))
        else:
            c = cfg.as_ctypes()
            self._chk(self.L.ksfd_set_stage_params(self.h, int(stage), C.byref(c)))

    def set_source(self, field, src, stage=-1):
        if src is None:
            self._chk(self.L.ksfd_set_source(self.h, stage, field, None, SOA))
        else:
            s = self._vec(src, self.nlocal // self.F)
            self._chk(self.L.ksfd_set_source(self.h, stage, field, _dp(s), SOA))

    # ---- operators
    def rhs(self, u=None, t=0.0, layout=SOA):
        out = np.empty(self.nlocal)
        up = _dp(self._vec(u)) if u is not None else None
        self._chk(self.L.ksfd_rhs(self.h, float(t), up, _dp(out), layout))
        return out

    def jvp(self, v, u=None, layout=SOA):
        out = np.empty(self.nlocal)
        v = self._vec(v)
        up = _dp(self._vec(u)) if u is not None else None
        self._chk(self.L.ksfd_jvp(self.h, up, _dp(v), _dp(out), layout))
        return out

    def velocity(self, u=None):
        n = self.cfg.dim * (self.nlocal // self.F)
        out = np.empty(n)
        up = _dp(self._vec(u)) if u is not None else None
        self._chk(self.L.ksfd_velocity(self.h, up, _dp(out), SOA))
        return out

    def velocity_max(self):
        v = np.zeros(3)
        self._chk(self.L.ksfd_velocity_max(self.h, _dp(v)))
        return v

    # ---- outer-loop helpers
    def groom(self):
        self._chk(self.L.ksfd_groom(self.h))

    def count_worms(self):
        t = C.c_double()
        self._chk(self.L.ksfd_count_worms(self.h, C.byref(t)))
        return t.value

    def scale_rho(self, f):
        self._chk(self.L.ksfd_scale_rho(self.h, float(f)))

    def mul_rho(self, factor):
        f = self._vec(factor, self.nlocal // self.F)
        self._chk(self.L.ksfd_mul_rho(self.h, _dp(f)))

    def snapshot_begin(self, layout=SOA):
        """start an asynchronous copy of the state to pinned host memory; returns the slot to pass to snapshot_wait"""
        slot = C.c_int32()
        self._chk(self.L.ksfd_snapshot_begin(self.h, layout, C.byref(slot)))
        return slot.value

    def snapshot_wait(self, slot):
        """numpy view (no copy) of the pinned buffer; valid until the second-next snapshot_begin"""
        ptr = C.POINTER(C.c_double)()
        self._chk(self.L.ksfd_snapshot_wait(self.h, slot, C.byref(ptr)))
        return np.ctypeslib.as_array(ptr, shape=(self.nlocal,))

    def checkpoint(self):
        """device-side copy of the state + the solver's step-to-step memory (one slot)"""
        self._chk(self.L.ksfd_checkpoint(self.h, 0))

    def restore(self):
        self._chk(self.L.ksfd_checkpoint(self.h, 1))

    def set_state_random(self, z_coarse, rho0=9000.0):
        """start_values on the device: z_coarse indexed [i,j,k] (x first) on the global coarse grid (ksfdsolver2.py:580-639)"""
        z = np.asarray(z_coarse, dtype=np.float64)
        nc = (C.c_int64 * 3)(*(list(z.shape) + [1, 1, 1])[:3])
        zf = np.ascontiguousarray(z.ravel(order='F'))
        self._chk(self.L.ksfd_set_state_random(self.h, nc, _dp(zf), float(rho0)))

    def jacobian_csr(self):
        """Assembled df/du at the resident state: (rowptr, col, val), local rows, global columns, unknown = F*point + dof
        (the reference's Vec ordering; KSFD/ksfdsym.py:814-886 + ksfdMat.pyx:55-180)."""
        nr, nnz = C.c_int64(), C.c_int64()
        self._chk(self.L.ksfd_jacobian_nnz(self.h, C.byref(nr), C.byref(nnz)))
        rowptr = np.empty(nr.value + 1, dtype=np.int64)
        col = np.empty(nnz.value, dtype=np.int64)
        val = np.empty(nnz.value)
        ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int64))
        self._chk(self.L.ksfd_jacobian_csr(self.h, ip(rowptr), ip(col), _dp(val)))
        return rowptr, col, val

    # ---- step
    def step(self, t, h, opts=None, raise_on_error=True):
        """One TS.step().  Returns (t_new, h_next, StepStats, rc)."""
        opts = opts or default_step_opts()
        tt, hh, st = C.c_double(t), C.c_double(h), StepStats()
        rc = self.L.ksfd_step(self.h, C.byref(tt), C.byref(hh), C.byref(opts), C.byref(st))
        if rc and raise_on_error:
            self._chk(rc)
        return tt.value, hh.value, st, rc

    def last_error(self):
        return (self.L.ksfd_last_error(self.h) or b'').decode()

    def last_error_vector(self, layout=SOA):
        out = np.empty(self.nlocal)
        self._chk(self.L.ksfd_get_last_error_vector(self.h, _dp(out), layout))
        return out

    # ---- measurement
    def set_profiling(self, on=True, only=None):
        """only: kernel class name ('jvp', ...) -- time just that class (cheaper: each event pair costs device time)"""
        code = int(bool(on))
        if on and only is not None:
            names = [self.L.ksfd_kernel_class_name(i).decode() for i in range(NKCLASS)]
            code = 2 + names.index(only)
        self._chk(self.L.ksfd_set_profiling(self.h, code))

    def profile(self, reset=False):
        p = Profile()
        self._chk(self.L.ksfd_get_profile(self.h, C.byref(p), int(reset)))
        names = [self.L.ksfd_kernel_class_name(i).decode() for i in range(NKCLASS)]
        return {n: dict(ms=p.ms[i], bytes=p.bytes[i], launches=p.launches[i], alg_bytes=p.alg_bytes[i]) for i, n in enumerate(names)}

    def synchronize(self):
        self._chk(self.L.ksfd_synchronize(self.h))

    def bench_kernel(self, cls, reps=20):
        ms, by = C.c_double(), C.c_double()
        self._chk(self.L.ksfd_bench_kernel(self.h, cls, reps, C.byref(ms), C.byref(by)))
        return ms.value, by.value

    def set_mg_params(self, nu=0, ncoarse_max=0, power_its=0, ratio=0.0, coarse_tol=0.0):
        self._chk(self.L.ksfd_set_mg_params(self.h, nu, ncoarse_max, power_its, ratio, coarse_tol))

    def set_poly_params(self, max_degree, target=0.0, mg_threshold=0.0):
        self._chk(self.L.ksfd_set_poly_params(self.h, int(max_degree), float(target), float(mg_threshold)))

    def spectral_apply(self, shift, v, layout=SOA):
        """z = (shift*I - J0)^-1 v, J0 = constant-coefficient part of the Jacobian at the resident state (test entry)"""
        out = np.empty(self.nlocal)
        v = self._vec(v)
        self._chk(self.L.ksfd_spectral_apply(self.h, float(shift), _dp(v), _dp(out), layout))
        return out

    def set_spectral_params(self, from_stiffness=0.0, enable=-1):
        self._chk(self.L.ksfd_set_spectral_params(self.h, float(from_stiffness), int(enable)))

    def set_tuning(self, use_fused=-1, yseg=0, yseg_jvp=None):
        """use_fused: bit0 = fused 2-D kernels, bit1 = recompute (non-frozen) Jacobian action."""
        self._chk(self.L.ksfd_set_tuning(self.h, use_fused, yseg, yseg if yseg_jvp is None else yseg_jvp))
