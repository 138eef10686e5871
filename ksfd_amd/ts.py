"""Host-side counterpart of the reference's time-stepper objects, with the same names and call
signatures, driving libksfd_hip.so instead of petsc4py:

    Derivatives   <- KSFD.Derivatives        (KSFD/ksfdsym.py:145-1209): dfdt / Jacobian action / velocity / groom
    KSFDTS        <- KSFD.ksfdts.KSFDTS      (KSFD/ksfdts.py:53-497):   solve loop, monitors, CFL, noise
    implicitTS    <- KSFD.ksfdts.implicitTS  (KSFD/ksfdts.py:500-561)

The loop order of KSFDTS.solve (KSFD/ksfdts.py:202-228) is kept: groom -> step -> noise -> CFL -> monitors.
The state lives on the GPU; monitors get a Vec-like object whose `.array` (PETSc dof-fastest layout,
KSFD/ksfdgrid.py:9-58) is fetched from the device only when a monitor touches it.
"""
import gc
from datetime import datetime

import numpy as np

from . import lib as klib
from .layout import PETSC, SOA, cijk_to_soa
from .options import SpatialExpression, grid_coords


class _Comm:
    def __init__(self, rank=0, size=1):
        self.rank, self.size = rank, size


class DeviceVec:
    """What monitors see as `u` (petsc4py Vec subset: .array, .assemble(), .copy())."""

    def __init__(self, ks):
        self._ks = ks
        self._cache = None

    def invalidate(self):
        self._cache = None

    @property
    def array(self):
        if self._cache is None:
            self._cache = self._ks.get_state(PETSC)
        return self._cache

    @array.setter
    def array(self, a):
        self._ks.set_state(np.asarray(a, dtype=np.float64), PETSC)
        self._cache = None

    def assemble(self):
        pass

    def duplicate(self):
        return HostVec(np.zeros(self._ks.nlocal))

    def copy(self, other=None):
        if other is None:
            return HostVec(self.array.copy())
        other.array[:] = self.array
        return other


class HostVec:
    def __init__(self, a):
        self.array = a

    def assemble(self):
        pass

    def setUp(self):
        pass

    def destroy(self):
        pass


class LocalGrid:
    """The Grid attributes monitors / TimeSeries read (KSFD/ksfdgrid.py:149-177), for this rank's slab."""

    def __init__(self, cfg, slow_range, rank=0, size=1):
        self.dim = cfg.dim
        self.dof = cfg.F
        self.nps = np.array(cfg.n[:cfg.dim])
        self.bounds = np.array(cfg.L[:cfg.dim])
        self.spacing = self.bounds / self.nps
        self.stencil_width = 2
        self.width, self.height, self.depth = cfg.L
        self.nx, self.ny, self.nz = cfg.n
        ranges = [(0, int(cfg.n[a])) for a in range(cfg.dim)]
        ranges[cfg.dim - 1] = tuple(int(x) for x in slow_range)
        self.ranges = self._ranges = tuple(ranges)
        self.Slshape = tuple(r[1] - r[0] for r in ranges)
        self.Vlshape = (self.dof,) + self.Slshape
        self.globalSshape = tuple(int(x) for x in cfg.n[:cfg.dim])
        self.globalVshape = (self.dof,) + self.globalSshape
        self.comm = _Comm(rank, size)

    def cleanup(self):
        pass


class Derivatives:
    """Operators on the device (KSFD/ksfdsym.py:902-940, 814-886, 1188-1209, 888-900)."""

    def __init__(self, ps, cfg, sources=None, ks=None, dist=None):
        self.ps = ps
        self.cfg = cfg
        self.ks = ks if ks is not None else klib.KSFDHip(cfg, dist)
        self.grid = LocalGrid(cfg, self.ks.slow_range, rank=dist.rank if dist else 0, size=dist.size if dist else 1)
        self.sources = sources if sources is not None else [SpatialExpression(ps, '0.0') for _ in range(cfg.F)]
        self.u0 = DeviceVec(self.ks)
        self.dim = cfg.dim
        self._coords = None
        self._src_static_done = False

    # -- sources(t) evaluated on the host (KSFD/ksfdsym.py:930-936) and uploaded per stage time
    def _local_coords(self):
        if self._coords is None:
            cs = grid_coords(self.cfg)
            lo, hi = self.ks.slow_range
            a = self.cfg.dim - 1
            sl = [slice(None)] * self.cfg.dim
            sl[a] = slice(lo, hi)
            cs[a] = cs[a][tuple(sl)]
            self._coords = cs
        return self._coords

    @staticmethod
    def _src_is_zero(s):
        if hasattr(s, 'is_zero'):
            return s.is_zero()
        e = getattr(s, 'expression', None)              # the reference's SpatialExpression
        return bool(e is not None and getattr(e, 'is_zero', False))

    def _src_eval(self, s, t):
        try:
            return s(t, self._local_coords())           # ksfd_amd.options.SpatialExpression
        except TypeError:
            return s(t)                                 # KSFD.SpatialExpression.__call__(t) (ksfdsym.py:1620-1655)

    def has_sources(self):
        return any(not self._src_is_zero(s) for s in self.sources)

    def upload_sources(self, times):
        """times: the 4 stage times (or one time for all stages)."""
        if not self.has_sources():
            return
        shape = self.grid.Slshape
        for c, s in enumerate(self.sources):
            if self._src_is_zero(s):
                continue
            for i, t in enumerate(times):
                v = np.broadcast_to(self._src_eval(s, t), shape)
                self.ks.set_source(c, v.ravel(order='F'), stage=i if len(times) > 1 else -1)

    def groom(self, farr):
        """in-place clamp of an array indexed [c, ...] (numpy path kept for host arrays)"""
        rhomin, Umin = self.cfg.rhomin, self.cfg.Umin
        farr[0] = np.maximum(farr[0], rhomin)
        farr[0][np.isnan(farr[0])] = rhomin
        farr[1:] = np.maximum(farr[1:], Umin)
        farr[1:][np.isnan(farr[1:])] = Umin
        return farr

    def dfdt(self, fvec=None, t=None, out=None):
        t = self.ps.t0 if t is None else t
        self.upload_sources([t])
        a = self.ks.rhs(None if fvec is None else fvec.array, t=t, layout=PETSC)
        if out is not None:
            out.array[:] = a
            return out
        return HostVec(a)

    def jvp(self, fvec, vvec):
        return HostVec(self.ks.jvp(vvec.array, None if fvec is None else fvec.array, layout=PETSC))

    def velocity(self, fvec=None, t=None, out=None):
        v = self.ks.velocity(None if fvec is None else cijk_to_soa(fvec.array.reshape(self.grid.Vlshape, order='F')))
        n = v.size // self.dim
        res = np.stack([v[a * n:(a + 1) * n].reshape(self.grid.Slshape, order='F') for a in range(self.dim)])
        if isinstance(out, np.ndarray):
            out[:] = res
            return out
        return res


class KSFDTS:
    """Base class for the HIP time-steppers (mirror of KSFD/ksfdts.py:53-497)."""

    default_hmin = 1e-20

    def __init__(self, derivs, t0=0.0, dt=0.001, tmax=20, maxsteps=100, rtol=1e-5, atol=1e-5, restart=True,
                 hmin=None, comm=None, opts=None, rng=None):
        self.derivs = derivs
        self.ks = derivs.ks
        self.comm = comm if comm is not None else derivs.grid.comm
        self.t0, self.tmax, self.maxsteps = float(t0), float(tmax), int(maxsteps)
        self.rtol, self.atol = float(rtol), float(atol)
        self.restart = restart
        self.hmin = float(hmin) if hmin else self.default_hmin
        self.opts = opts if opts is not None else klib.default_step_opts()
        self.opts.rtol, self.opts.atol = self.rtol, self.atol
        self.history = []
        self.u = derivs.u0
        self._t, self._h, self._k = self.t0, float(dt), 0
        self.diverged = False
        self.reason = ''
        self._snes_failures = 0
        self._monitors = []
        self.rng = rng
        self.last_stats = None
        self.stats_log = []
        self.CFL_maxh = self.CFL_step(self.u)

    # ---- petsc4py.TS subset used by the reference's main() and monitors
    def setMonitor(self, fn, args=None, kargs=None):
        self._monitors.append((fn, tuple(args or ()), dict(kargs or {})))

    def monitor(self, k, t, u):
        for fn, a, kw in self._monitors:
            fn(self, k, t, u, *a, **kw)

    def getStepNumber(self):
        return self._k

    def getTimeStep(self):
        return self._h

    def setTimeStep(self, h):
        self._h = float(h)

    def getTime(self):
        return self._t

    def setTime(self, t):
        self._t = float(t)

    def getSolution(self):
        return self.u

    def setSolution(self, u):
        if u is not self.u:
            self.u.array = u.array

    def getMaxTime(self):
        return self.tmax

    def getMaxSteps(self):
        return self.maxsteps

    def setMaxSteps(self, n):
        self.maxsteps = int(n)

    def getSNESFailures(self):
        return self._snes_failures

    def setFromOptions(self):
        pass

    ASUM = (0.0, 8.7173304301691801e-01, 8.4457060015369423e-01 - 1.1299064236484185e-01, 1.0)     # stage abscissae of RA34PW2

    def _time_dependent(self):
        """names of PDE coefficients that are expressions of t (ksfdsolver2.py:149-173); usually none"""
        ps = self.derivs.ps
        if not hasattr(ps, 'time_dependent') or not hasattr(ps, 'problem_config'):
            return []
        if not hasattr(self, '_td'):
            self._td = [k for k in ps.time_dependent() if k not in ('variance_timing_function', 't')]
        return self._td

    def _refresh_time_dependent_params(self, h):
        """The reference hands ps.values(t) to its ufuncs on every call: the Jacobian sees t_n (implicitIJ, KSFD/ksfdts.py:
        598-640), each stage RHS its own stage time t_n + ASum_i h (implicitIF, :563-596; KSFD/ksfdsym.py:1303-1312,
        1430-1439).  One small table per stage crosses the ABI (ksfd_update_params / ksfd_set_stage_params)."""
        ps = self.derivs.ps
        self.ks.update_params(ps.problem_config(self._t))
        for i, a in enumerate(self.ASUM):
            self.ks.set_stage_params(i, ps.problem_config(self._t + a * h))

    # ---- one TS.step(): replaces super().step() at KSFD/ksfdts.py:211
    def step(self):
        d = self.derivs
        tdep = bool(self._time_dependent())
        if d.has_sources() or tdep:
            # stage-time data (sources, time-dependent coefficients) belong to the stage times t + ASum_i*h of THIS attempt:
            # one attempt per call (opts.reserved bit 1), the reject loop runs here; bit 2 carries "previous attempt rejected"
            base = self.opts.reserved & 1
            prev_rejected = False
            limit = self.opts.max_reject if self.opts.max_reject >= 0 else 1 << 30
            rejections = 0
            while True:
                if tdep:
                    self._refresh_time_dependent_params(self._h)
                if d.has_sources():
                    d.upload_sources([self._t + a * self._h for a in self.ASUM])
                self.opts.reserved = base | 2 | (4 if prev_rejected else 0)
                t, h, st, rc = self.ks.step(self._t, self._h, self.opts, raise_on_error=False)
                self._h = h
                if rc or st.accepted:
                    break
                prev_rejected = True
                rejections += 1
                if rejections > limit:
                    break
            self.opts.reserved = base
            st.rejections = rejections
        else:
            t, h, st, rc = self.ks.step(self._t, self._h, self.opts, raise_on_error=False)
        self.last_stats = st
        self.stats_log.append((st.accepted, st.rejections, st.linear_its, st.wrms, st.h_used, st.bytes))
        self.u.invalidate()
        if rc:
            self.reason = self.ks.last_error()
            if rc == klib.ELINEAR:
                self._snes_failures += 1          # setMaxSNESFailures(1), KSFD/ksfdts.py:135
            self.diverged = True
            return
        if st.accepted:
            self._k += 1
            self._t, self._h = t, h
        else:
            self.diverged = True
            self.reason = 'step rejected repeatedly'

    def solve(self, u=None):
        """KSFD/ksfdts.py:170-229"""
        if u is not None:
            self.setSolution(u)
        u = self.u
        tmax, kmax = self.getMaxTime(), self.getMaxSteps()
        k, h = self.getStepNumber(), self.getTimeStep()
        self.CFL_check()
        t = self.getTime()
        Nworms = self.count_worms(u)
        p0 = self.derivs.ps.params0
        self.lastvart = float(p0['lastvart']) if 'lastvart' in p0 else t        # KSFD/ksfdts.py:193-196 (solver.main sets it on --resume/--restart)
        cw = p0.get('conserve_worms', False)
        conserve_worms = False if cw == 'False' else bool(cw)
        self.monitor(k, t, u)
        while (not self.diverged) and k < kmax and t <= tmax and h >= self.hmin:
            self.groom(u)                       # also done on the device inside ksfd_step
            self.step()
            if k % 20 == 0:
                gc.collect()
            k, h, t, u = self.getStepNumber(), self.getTimeStep(), self.getTime(), self.getSolution()
            if self.diverged:
                break
            dt = t - self.lastvart
            if self.is_noise_time(t, self.lastvart):
                u = self.add_variance(u, dt)
                if conserve_worms:
                    u = self.conserve_worms(u, Nworms)
                self.lastvart = t
            self.CFL_check()
            self.monitor(k, t, u)

    # ---- KSFD/ksfdts.py:231-284
    def groom(self, u):
        self.ks.groom()
        self.u.invalidate()
        return u

    def count_worms(self, u):
        return self.ks.count_worms()

    def conserve_worms(self, u, Nworms):
        self.ks.scale_rho(Nworms / self.ks.count_worms())
        self.u.invalidate()
        return u

    def is_noise_time(self, t, lastvart):
        vrate = self.derivs.ps.values(t)['variance_rate']
        if not vrate or vrate <= 0.0:
            return False
        flast = self.derivs.ps.values(lastvart)['variance_timing_function']
        fnow = self.derivs.ps.values(t)['variance_timing_function']
        return fnow - flast >= 1.0

    def add_variance(self, u, dt):
        t = self.getTime()
        vrate = self.derivs.ps.values(t)['variance_rate']
        if not vrate or vrate <= 0.0:
            return u
        sd = np.sqrt(vrate * dt)
        if self.rng is None:
            from .initial import reference_rng
            self.rng = reference_rng(rank=self.comm.rank, size=self.comm.size)
        stn_sample = self.rng.normal(size=self.derivs.grid.Slshape)     # same stream/shape as ksfdts.py:279-280
        self.ks.mul_rho(np.exp(sd * stn_sample).ravel(order='F'))
        self.u.invalidate()
        return u

    # ---- KSFD/ksfdts.py:287-319
    def CFL_check(self):
        h, t = self.getTimeStep(), self.getTime()
        self.CFL_maxh = self.CFL_step(self.u, t)
        safety = self.derivs.ps.values(t)['CFL_safety_factor']
        if safety > 0.0:
            maxh = safety * self.CFL_maxh
            if h > maxh:
                self.setTimeStep(maxh)

    def CFL_step(self, u, t=None):
        vmax = self.ks.velocity_max()[:self.derivs.dim]         # already max over ranks
        sw = self.derivs.grid.stencil_width
        hmaxs = [float('inf') if v == 0.0 else s * sw / v for v, s in zip(vmax, self.derivs.grid.spacing)]
        return float(np.min(hmaxs))

    def cleanup(self):
        self.ks.close()

    # ---- monitors (KSFD/ksfdts.py:337-497)
    def printMonitor(self, ts, k, t, u):
        if self.comm.rank == 0:
            h = ts.getTimeStep()
            if hasattr(self, 'lastt'):
                out = "clock: %s, step %3d t=%8.3g dt=%8.3g h=%8.3g" % (
                    datetime.now().strftime('%H:%M:%S'), k, t, t - self.lastt, h)
            else:
                out = "clock: %s, step %3d t=%8.3g h=%8.3g" % (datetime.now().strftime('%H:%M:%S'), k, t, h)
            if hasattr(self, 'CFL_maxh'):
                out += ' CFL=%8.3g' % (self.CFL_maxh)
            if self.last_stats is not None:
                out += ' its=%d rej=%d' % (self.last_stats.linear_its, self.last_stats.rejections)
            print(out, flush=True)
            self.lastt = t

    def historyMonitor(self, ts, k, t, u):
        self.history.append(dict(step=k, h=ts.getTimeStep(), t=t, u=u.array.copy()))

    def checkpointMonitor(self, ts, k, t, u, prefix, mpiok=False):
        """One single-point series per step, <prefix>_<k>_s<size>r<rank> (KSFD/ksfdts.py:370-451; no zip option)."""
        from .timeseries import TimeSeries
        cpf = TimeSeries(prefix + '_' + str(k) + '_', self.derivs.grid, mode='w')
        cpf.set_dt(float(ts.getTimeStep()))
        cpf.info['lastvart'] = float(getattr(self, 'lastvart', t))
        cpf.store(u, t, k=k)
        cpf.close()

    def makeSaveMonitor(self, timeseries):
        self.timeseries = timeseries

        def closeSaveMonitor():
            pass

        def saveMonitor(ts, k, t, u):
            if hasattr(self.timeseries, 'tsFile') and not self.timeseries.tsFile:
                self.timeseries.reopen()
            self.timeseries.store(u, t, k=k)
            if hasattr(self.timeseries, 'set_dt'):
                self.timeseries.set_dt(float(ts.getTimeStep()))
            if hasattr(self.timeseries, 'info'):
                # (superset of the reference, whose save monitor records dt only and leaves lastvart to the checkpoint monitor,
                #  KSFD/ksfdts.py:420-422: with it a run resumed from a --save series injects noise on schedule too)
                self.timeseries.info['lastvart'] = float(getattr(self, 'lastvart', t))
            if hasattr(self.timeseries, 'temp_close'):
                self.timeseries.temp_close()

        return (saveMonitor, closeSaveMonitor)


class implicitTS(KSFDTS):
    """Fully implicit timestepper (KSFD/ksfdts.py:500-561): IFunction = udot - f(u), IJacobian = shift*I - J,
    both applied on the device inside ksfd_step."""
    pass
