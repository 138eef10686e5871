#!/usr/bin/env python3
"""bench.py -- grid-point-updates/s of the implicit Keller-Segel step on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one implicit time step (KSFDTS.solve loop body, KSFD/ksfdts.py:207-228: groom -> 4-stage
ROSW RA34PW2 step with matrix-free GMRES -> CFL velocity check) of the BASELINE.json headline config:
2-D 4096^2, one ligand, fp64, synthetic random-perturbation initial data, state resident
in HBM.  The SAME global grid is slab-decomposed over the N GPUs (strong scaling).

One JSON line on rank 0, with `roofline` for the dominant kernel (algorithmic bytes per launch / HIP-event
time per launch measured on the library's compute stream during the timed region) and `cpu_baseline`
(the oracle's restatement of the same step, timed on the host cores on a bounded sub-grid sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def build_problem(n, nlig, spacing_ref=4.0 / 1536, dim=2):
    """options84-style physics (options84:20-46) on an n^dim grid with the reference run's spacing
    (width = 4*(n/1536))."""
    from ksfd_amd.config import ProblemConfig
    L = n * spacing_ref
    return ProblemConfig.standard(dim, (n,) * dim, L=(L,) * dim, nlig=nlig)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--grid', dest='n', type=int, default=4096, help='points per axis')
    ap.add_argument('--nlig', type=int, default=1)
    ap.add_argument('--dim', type=int, default=2, help='2 (headline) or 3 (BASELINE configs[4]-style, generic kernels)')
    ap.add_argument('--dt', type=float, default=0.01, help='first trial step (the controller adapts from here)')
    ap.add_argument('--fixed-h', type=float, default=0.0, help='>0: -ts_adapt_type none with this step')
    ap.add_argument('--ksp-rtol', type=float, default=1e-6,
                    help='GMRES relative residual; 1e-6 keeps the fields within ~1e-10 rel-L2 of a 1e-12 solve')
    ap.add_argument('--transport', default=os.environ.get('KSFD_TRANSPORT', 'auto'), choices=['auto', 'rccl', 'host'])
    ap.add_argument('--dist-backend', default='nccl', choices=['nccl', 'gloo'],
                    help='torch.distributed backend for launch/timing; gloo + --transport host lets several ranks share one GPU (rehearsal)')
    ap.add_argument('--share-gpu', action='store_true', help='rehearsal: every rank uses device 0')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-sample-n', type=int, default=3072)
    ap.add_argument('--yseg', type=int, default=0)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from ksfd_amd import lib as klib
    from ksfd_amd.dist import open_handle

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if rank == 0:
            print('bench.py: --gpus %d but WORLD_SIZE=%d; launch with torch.distributed.run' % (args.gpus, world),
                  file=sys.stderr)
        sys.exit(2)
    host_group = None
    dev = 0 if (world == 1 or args.share_gpu) else local_rank
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        torch.cuda.set_device(dev)
        if args.dist_backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', dev))
            host_group = dist.new_group(backend='gloo')
        else:
            dist.init_process_group('gloo')

    cfg = build_problem(args.n, args.nlig, dim=args.dim)
    ks, keep = open_handle(cfg, rank, world, dev, transport=args.transport, group=None, host_group=host_group)
    if args.yseg:
        ks.set_tuning(yseg=args.yseg)
    if os.environ.get('KSFD_POLY_DEG'):
        ks.set_poly_params(int(os.environ['KSFD_POLY_DEG']), float(os.environ.get('KSFD_POLY_TARGET', '0')))
    if os.environ.get('KSFD_TUNE'):                      # A/B switches of ksfd_set_tuning (tools/async_bench.py)
        ks.set_tuning(use_fused=int(os.environ['KSFD_TUNE']))
    # synthetic start values of SURVEY.md 8d, interpolated on the device slab by slab from the global coarse samples
    # (seed 793817931, n/4 coarse normal noise sigma=90 around rho=9000, U = rho*s/gamma): ksfd_set_state_random
    from ksfd_amd.initial import reference_rng
    ks.set_state_random(reference_rng().normal(size=tuple(max(1, n // 4) for n in cfg.n[:cfg.dim])) * 90.0, 9000.0)

    if args.fixed_h > 0:
        opts = klib.default_step_opts(adapt=0, atol=0.01, rtol=1e-6, ksp_rtol=args.ksp_rtol)
        h = args.fixed_h
    else:
        opts = klib.default_step_opts(adapt=1, atol=0.01, rtol=1e-6, ksp_rtol=args.ksp_rtol)   # options84:18-19
        h = args.dt
    t = 0.0
    spacing = cfg.spacing

    def one_step(t, h):
        t, h, st, rc = ks.step(t, h, opts)                   # groom + ROSW/GMRES step (+ rejections)
        vmax = ks.velocity_max()                              # CFL_check, KSFD/ksfdts.py:287-319
        cfl = min(s * 2 / v if v > 0 else float('inf') for s, v in zip(spacing, vmax[:cfg.dim]))
        return t, h, st, cfl

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        ks.synchronize()

    # Warm-up steps run with a HIP-event pair around every launch: per-class table + which class dominates.  Every
    # event pair costs ~8 us of device time, so the timed region keeps events only around the dominant class (the
    # one the roofline object describes); the launch/byte counters of the other classes keep running.
    ks.set_profiling(True)
    ks.profile(reset=True)
    for _ in range(args.warmup):
        t, h, st, cfl = one_step(t, h)
    warm = ks.profile(reset=True)
    skip = ('halo', 'reduce', 'misc')                       # transport / tiny kernels are not roofline material
    dom = max((k for k in warm if k not in skip), key=lambda k: warm[k]['ms']) if args.warmup > 0 else None
    if world > 1 and dom is not None:                       # every rank must time the same class
        names = sorted(warm)
        pick = torch.tensor([names.index(dom)], dtype=torch.int64, device='cuda' if args.dist_backend == 'nccl' else 'cpu')
        dist.broadcast(pick, 0)
        dom = names[int(pick.item())]
    ks.set_profiling(not os.environ.get('KSFD_BENCH_NOPROF'), only=dom)      # env knob: A/B the cost of the events
    ks.profile(reset=True)
    barrier()
    t_start = time.perf_counter()
    its, rej, hs, nbytes = 0, 0, [], 0.0
    for _ in range(args.steps):
        t, h, st, cfl = one_step(t, h)
        its += st.linear_its
        rej += st.rejections
        hs.append(st.h_used)
        nbytes += st.bytes
    barrier()
    elapsed = time.perf_counter() - t_start
    prof = ks.profile()
    ks.set_profiling(False)
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device='cuda' if args.dist_backend == 'nccl' else 'cpu')
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    if rank == 0:
        N = cfg.N
        value = N * args.steps / elapsed
        # dominant kernel = largest share of device time in the timed region
        if dom is None:
            dom = max((k for k in prof if k not in skip), key=lambda k: prof[k]['ms'])
        d = prof[dom]
        per_launch_bytes = d['bytes'] / max(d['launches'], 1)
        per_launch_ms = d['ms'] / max(d['launches'], 1)
        achieved = per_launch_bytes / (per_launch_ms * 1e-3) / 1e9 if per_launch_ms > 0 else 0.0
        table = warm if args.warmup > 0 else prof             # all-class timings: warm-up steps (see above)
        kern = {k: dict(ms=round(v['ms'], 3), launches=int(v['launches']),
                        GBs=round(v['bytes'] / (v['ms'] * 1e-3) / 1e9, 1) if v['ms'] > 0 else None)
                for k, v in table.items() if v['launches']}
        # the committed PMC summary was taken on the default workload only
        default_workload = world == 1 and args.n == 4096 and args.dim == 2 and args.nlig == 1 and args.fixed_h == 0
        traffic, traffic_src = pmc_traffic(dom) if default_workload else (None, None)
        out = {
            'metric': 'grid-point-updates/sec (implicit step)', 'value': value, 'unit': 'grid-point-updates/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps,
            'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': '%dD %s %d-ligand Keller-Segel, ROSW RA34PW2 + matrix-free GMRES(30,CGS2), '
                                   'options84 spacing/physics, %s' % (args.dim, 'x'.join([str(args.n)] * args.dim), args.nlig,
                                                                      'fixed h=%g' % args.fixed_h if args.fixed_h > 0 else
                                                                      'TSAdaptBasic rtol=1e-6 atol=0.01 from dt=%g' % args.dt),
                       'grid': [args.n] * args.dim, 'fields': cfg.F, 'ksp_rtol': args.ksp_rtol,
                       'h_mean': float(np.mean(hs)), 'gmres_its_per_step': its / args.steps, 'rejections': rej,
                       't_end': t, 'parallelism': 'slab%d' % world, 'preconditioner': 'Chebyshev polynomial p(A), fp32 storage of its temporaries/coefficient copy (fp64 arithmetic; Krylov vectors, A z_j, solution fp64)' if not (int(os.environ.get('KSFD_TUNE', '1')) & 512) else 'Chebyshev polynomial p(A), fp64',
                       'transport': type(keep).__name__ if keep is not None else 'none'},
            'roofline': {'bound': 'hbm', 'kernel': dom, 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic, 'traffic_source': traffic_src,
                         'bytes_per_launch': per_launch_bytes, 'ms_per_launch': per_launch_ms,
                         'step_algorithmic_GBs': nbytes / elapsed / 1e9,
                         'step_frac_of_peak': nbytes / elapsed / 1e9 / HBM_PEAK_GBS},
            'kernels': kern, 'kernels_region': 'warmup steps (events on every launch)' if args.warmup > 0 else 'timed steps',
        }
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(args, float(np.mean(hs)))
        print(json.dumps(out), flush=True)
    ks.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def pmc_traffic(kernel_class):
    """HBM bytes per launch of this kernel class from the newest committed PMC summary (profiles/*_pmc.json,
    produced by tools/prof.sh + tools/summarize_prof.py from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
    passes of this same command, gfx950 FETCH_SIZE x2 correction applied).  None if no summary is present."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_pmc.json')))
    if not files:
        return None, None
    try:
        c = json.load(open(files[-1]))['classes'].get(kernel_class)
        return (c['hbm_bytes_per_launch'], os.path.basename(files[-1])) if c else (None, None)
    except Exception:
        return None, None


def usable_cores():
    """Host cores this process may really use: cgroup quota if there is one, else the affinity mask,
    capped at 16 (the GPU box's CPU share per GPU; its affinity mask shows all 256 hardware threads)."""
    n = len(os.sched_getaffinity(0))
    try:
        q, p = open('/sys/fs/cgroup/cpu.max').read().split()
        if q != 'max':
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    return min(n, 16)


def cpu_baseline(args, h):
    """The oracle's restatement of the same step (kind 'port'), OpenMP over the host cores this process may
    use, on a bounded sample: ONE step at the GPU run's mean h on an m x m sub-grid with the same spacing,
    physics and initial-data statistics."""
    from ksfd_amd.initial import start_values
    from oracle import ko
    m = args.cpu_sample_n
    if args.dim == 3:
        m = min(m, 64)
    cfg = build_problem(m, args.nlig, dim=args.dim)
    u = start_values(cfg)
    cores = usable_cores()
    ko.set_threads(cores)
    o = ko.Oracle(cfg)
    t0 = time.perf_counter()
    un, err, wr, its = o.rosw_step(u, h, 0.01, 1e-6, solver='gmres', ksp_rtol=args.ksp_rtol, restart=30, maxit=2000)
    dt = time.perf_counter() - t0
    ko.set_threads(1)
    return {'value': cfg.N / dt, 'unit': 'grid-point-updates/s', 'cores': cores, 'kind': 'port',
            'sample': 'one ROSW+GMRES(30,CGS2) step at h=%.4g on a %s sub-grid (same spacing/physics/IC '
                      'statistics), %d GMRES its, %.1f s' % (h, 'x'.join([str(m)] * args.dim), its, dt)}


if __name__ == '__main__':
    main()
