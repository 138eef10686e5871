#!/usr/bin/env python3
"""`python -m ksfd_amd.solver @options ...` -- counterpart of the reference's ksfdsolver2.main
(ksfdsolver2.py:642-774) with the PETSc TS replaced by the HIP stepper.  Same command-line syntax."""
import os
import sys

import numpy as np

from . import options as opt
from .initial import reference_rng, smoothstep_interpolate
from .ts import Derivatives, implicitTS
from .timeseries import TimeSeries


def start_values(ps, cfg, rng):
    """ksfdsolver2.py:580-639: rho = rho0(x) + interpolated coarse noise; U = U0(x) if given else rho*s/gamma."""
    v = ps.values0
    dim, shape = cfg.dim, cfg.n[:cfg.dim]
    coarse = []
    for a, key in enumerate(('randgridnw', 'randgridnh', 'randgridnd')[:dim]):
        coarse.append(int(v[key]) if v[key] else max(1, shape[a] // 4))
    murho0 = float(v['Nworms']) / (cfg.L[0] ** dim)
    sigma = float(v['srho0'])
    if sigma == 0.0:
        noise = np.full(shape, murho0)
    else:
        z = rng.normal(size=tuple(coarse)) * sigma + murho0
        noise = smoothstep_interpolate(z, shape)
    coords = opt.grid_coords(cfg)
    rho0 = v['rho0']
    rho = np.broadcast_to(opt.SpatialExpression(ps, rho0)(ps.t0, coords), shape) + noise if rho0 else noise
    fields = [rho]
    names = ps.field_names()[1:]
    for l, name in enumerate(names):
        key = 'U0' + name[1:]
        U0 = v.get(key)
        if U0 is not None and U0 is not False and U0 != '':
            fields.append(np.broadcast_to(opt.SpatialExpression(ps, U0)(ps.t0, coords), shape) + 0.0)
        else:
            fields.append(rho * (cfg.lig_s[l] / cfg.lig_gamma[l]))
    return np.concatenate([f.ravel(order='F') for f in fields])          # SoA, x fastest


class _Rank:
    def __init__(self, rank, size):
        self.rank, self.size = rank, size


def open_ranks(cfg):
    """One process per GPU under torch.distributed.run (RANK/LOCAL_RANK/WORLD_SIZE), the counterpart of the reference's
    `mpiexec -n P` SPMD launch (ksfdsolver2.py:650-652).  Returns (KSFDHip, _Rank, keepalive).
    Environment: KSFD_TRANSPORT=auto|rccl|host, KSFD_DIST_BACKEND=nccl|gloo, KSFD_SHARE_GPU=1 (all ranks on device 0:
    rehearsal on a one-GPU box)."""
    from . import lib as klib
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world == 1:
        return klib.KSFDHip(cfg), _Rank(0, 1), None
    import torch
    import torch.distributed as dist
    from .dist import open_handle
    rank, local = int(os.environ['RANK']), int(os.environ.get('LOCAL_RANK', '0'))
    dev = 0 if os.environ.get('KSFD_SHARE_GPU') else local
    backend = os.environ.get('KSFD_DIST_BACKEND', 'nccl')
    host_group = None
    torch.cuda.set_device(dev)
    if not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', dev))
        else:
            dist.init_process_group('gloo')
    if dist.get_backend() == 'nccl':
        host_group = dist.new_group(backend='gloo')
    default_tr = 'auto' if dist.get_backend() == 'nccl' else 'host'
    ks, keep = open_handle(cfg, rank, world, dev, transport=os.environ.get('KSFD_TRANSPORT', default_tr), host_group=host_group)
    return ks, _Rank(rank, world), keep


def main(*args):
    argv = list(args) if args else sys.argv
    cl = opt.parse_commandline(argv[1:])
    if cl.noperiodic:
        raise ValueError('--periodic=false not implemented (as in the reference, ksfdsolver2.py:657-661)')
    ps = opt.Params(cl)
    if cl.showparams:
        for k, val in ps.values0.items():
            print('%s=%s' % (k, val))
        return None
    cfg = ps.problem_config()
    sources = opt.decode_sources(cl.source, ps)
    ks, rk, keep = open_ranks(cfg)
    # initial data: the global field from the rank-0 stream on every rank (identical for any number of ranks), then
    # this rank's slab; noise injection later draws from the per-rank stream like the reference (ksfdrandom.py:44-49)
    rng = reference_rng(cl.seed)
    derivs = Derivatives(ps, cfg, sources, ks=ks, dist=rk if rk.size > 1 else None)
    from .layout import SOA, HDF5
    from .dist import local_slab
    v = ps.values0
    t_start, dt0, k0 = ps.t0, float(v['dt']), 0
    resuming = cl.resume or cl.restart
    if resuming:
        # ksfdsolver2.py:525-578: last point of the series; --resume keeps its time and dt, --restart starts at t0
        from .timeseries import read_last
        k_last, t_last, data, info, _ = read_last(resuming, size=rk.size, rank=rk.rank)
        derivs.ks.set_state(np.ascontiguousarray(data).ravel(), HDF5)
        if cl.resume:
            t_start = t_last
            if 'dt' not in ps.given and 'dt' in info:
                dt0 = float(info['dt'])
            # time of the last variance injection (resume_values, ksfdsolver2.py:554-561): explicit parameter, else what the
            # series recorded, else the resume time -- so that a resumed run injects noise when the uninterrupted one would
            if 'lastvart' not in ps.given:
                ps.params0['lastvart'] = float(info['lastvart']) if 'lastvart' in info else t_last
        elif 'lastvart' not in ps.given:
            ps.params0['lastvart'] = ps.t0                       # --restart: ksfdsolver2.py:562-567
    else:
        u0 = start_values(ps, cfg, rng)
        derivs.ks.set_state(local_slab(u0, cfg, rk.rank, rk.size) if rk.size > 1 else u0, SOA)
        if rk.size > 1:
            rng = reference_rng(cl.seed, rank=rk.rank, size=rk.size)
    ts = implicitTS(derivs, t0=t_start, dt=dt0, tmax=float(v['tmax']),
                    maxsteps=0 if cl.onestep else int(v['maxsteps']), rtol=float(v['rtol']), atol=float(v['atol']),
                    opts=opt.step_opts_from(ps, cl.petsc), rng=rng)
    ts.setMonitor(ts.printMonitor)
    tseries = None
    if cl.save:
        tseries = TimeSeries(cl.save, derivs.grid, mode='w', async_save=cl.async_save, save_every=cl.saveevery)
        tseries.set_dt(float(v['dt']))
        save, closer = ts.makeSaveMonitor(tseries)
        ts.setMonitor(save)
    if cl.check:
        ts.setMonitor(ts.checkpointMonitor, (cl.check,))
    try:
        ts.solve()
    finally:
        if tseries is not None:
            tseries.close()
    if ts.comm.rank == 0:
        print('SNES failures = ', ts.getSNESFailures())
    return ts


if __name__ == '__main__':
    t = main()
    if t is not None:
        t.cleanup()
