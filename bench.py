#!/usr/bin/env python3
"""bench.py -- grid-point-updates/s of the implicit Keller-Segel step on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1 without a launcher: this script starts `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
   --master-addr 127.0.0.1 ... bench.py ...` itself, before anything touches the GPU, and relays rank 0's line)

A "step" is one implicit time step (KSFDTS.solve loop body, KSFD/ksfdts.py:207-228: groom -> 4-stage ROSW RA34PW2 step,
every stage system solved matrix-free on the device (which solver ran is reported per step) -> CFL velocity check) of the
BASELINE.json headline config: 2-D 4096^2, one
ligand, fp64, options84 physics/spacing, synthetic random-perturbation initial data, state resident in
HBM.  The SAME global grid is slab-decomposed over the N GPUs (strong scaling).

What is timed (so that `value` does not depend on --steps/--warmup: the controller lets h grow 7 orders of magnitude
over a run, and a step at h = 1 costs several times a step at h = 1e-3):

  setup (untimed for `value`, reported as `ramp`): the adaptive run from the reference's dt0 = 1e-8 (options84:10) with
      TSAdaptBasic rtol = 1e-6, atol = 0.01 (options84:18-19) up to model time t* = --t-star (default 0.35).  The state,
      the proposed step and the solver's step-to-step memory at t* are checkpointed on the device.
  primary `value`: K steps walking cyclically through the PINNED WINDOW = the next --window (5) adaptive steps from that
      checkpoint (ksfd_checkpoint restore at every wrap-around: one device copy, inside the timed region).  Every
      window is the same work, so any K that is a multiple of the window gives the same number; W warm-up steps walk the
      same window first.  value = N_grid * K / wall.
  `fixed_h`: 100 steps at h = 1e-3 from the start values (SURVEY.md 8d's second run), its own rate.
  `aggregated`: the late phase of a production run, where the wall clock of a long run goes (aggregates have formed, the
      constant-coefficient inverse no longer contracts, the multigrid-preconditioned GMRES takes over): a 256^2 run with
      options81 spacing from dt0 to the aggregated state, tiled periodically to 2048^2 (an exact solution of the larger
      problem), then a few adaptive steps timed there.  Its own rate, never part of `value`.

One JSON line on rank 0, with `roofline` for the dominant kernel class of the timed region (HIP events on the library's
compute stream), `cpu_baseline` (the oracle's restatement of the same step with unpreconditioned GMRES, OpenMP over the host
cores, bounded sample) and `cpu_baseline_lu` (the REFERENCE's algorithm -- assembled Jacobian, one sparse LU per step, four
back-substitutions, options84:58-60 -- from the oracle's operators + scipy's SuperLU on a bounded sample).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--grid', dest='n', type=int, default=4096, help='points per axis')
    ap.add_argument('--nlig', type=int, default=1)
    ap.add_argument('--dim', type=int, default=2, help='2 (headline) or 3 (BASELINE configs[4]-style)')
    ap.add_argument('--dt0', type=float, default=1e-8, help='first trial step of the ramp (options84:10)')
    ap.add_argument('--t-star', type=float, default=0.35,
                    help='model time at which the pinned window starts (0.35: the window then spans h ~ 0.1-0.3, the regime BENCH_r01 measured)')
    ap.add_argument('--window', type=int, default=5, help='adaptive steps in the pinned window')
    ap.add_argument('--fixed-h', type=float, default=1e-3, help='step of the fixed-h leg (SURVEY.md 8d); 0 skips it')
    ap.add_argument('--fixed-steps', type=int, default=100)
    ap.add_argument('--ksp-rtol', type=float, default=0.0, help='GMRES relative residual; 0 = library default')
    ap.add_argument('--pc-type', type=int, default=-1, help='ksfd_step_opts.pc_type; -1 = library default (automatic)')
    ap.add_argument('--transport', default=os.environ.get('KSFD_TRANSPORT', 'auto'), choices=['auto', 'rccl', 'host'])
    ap.add_argument('--dist-backend', default='nccl', choices=['nccl', 'gloo'],
                    help='torch.distributed backend for launch/timing; gloo + --transport host lets several ranks share one GPU (rehearsal)')
    ap.add_argument('--share-gpu', action='store_true', help='rehearsal: every rank uses device 0')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-autotune', action='store_true', help='N > 1: keep the library\'s default stage solver instead of timing the window with and without the distributed spectral solver')
    ap.add_argument('--no-live-pmc', action='store_true', help='take roofline.traffic from the committed PMC summary instead of measuring it in two rocprofv3 child runs')
    ap.add_argument('--cpu-sample-n', type=int, default=1024)
    ap.add_argument('--cpu-lu-n', type=int, default=256, help='grid of the sparse-LU CPU baseline (0 skips it); SuperLU needs ~15 s at 256^2, > 80 s at 512^2')
    ap.add_argument('--agg-tile', type=int, default=256, help='grid of the run that produces the aggregated state (0 skips the leg)')
    ap.add_argument('--agg-n', type=int, default=2048, help='grid the aggregated state is tiled to')
    ap.add_argument('--agg-steps', type=int, default=5)
    ap.add_argument('--yseg', type=int, default=0)
    return ap.parse_args(argv)


def self_launch(args):
    """`python bench.py --gpus N` from a bare shell: start the N ranks as a CHILD process group (never exec from a process
    that has touched the GPU; this parent imports neither torch nor the library), relay output and exit code."""
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    return subprocess.call(cmd, env=env)


def build_problem(n, nlig, spacing_ref=4.0 / 1536, dim=2):
    """options84-style physics (options84:20-46) on an n^dim grid with the reference run's spacing
    (width = 4*(n/1536))."""
    from ksfd_amd.config import ProblemConfig
    L = n * spacing_ref
    return ProblemConfig.standard(dim, (n,) * dim, L=(L,) * dim, nlig=nlig)


PC_NAMES = {1: 'none', 2: 'multigrid', 4: 'polynomial', 8: 'spectral'}
SOLVER_TEXT = {
    'spectral': 'defect correction x += M^-1 (b - A x), M = constant-coefficient part of shift*I - J inverted by hand-written LDS FFTs; '
                'no Krylov vectors; stops on the true fp64 residual (last sweep of a solve applied on the measured contraction)',
    'none': 'unpreconditioned restarted GMRES(30)',
    'polynomial': 'flexible GMRES(30) with a Chebyshev polynomial preconditioner (degree <= 6)',
    'multigrid': 'GMRES(30) right-preconditioned with one geometric-multigrid V(2,2) cycle',
}
MIXED_PRECISION = ('fp64: state, stage vectors, right-hand sides, Jacobian action, residual norm and stopping test, step completion and '
                   'error norm; fp32: the FFT work array and arithmetic of M^-1 (spectral solver), the STORED residual that only M^-1 reads '
                   '(its norm is accumulated from the fp64 values before rounding), coefficient copy + Horner temporaries inside the '
                   'polynomial preconditioner, block-diagonal inverses of the multigrid smoother')


def main():
    args = parse_args()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if args.gpus > 1 and 'RANK' not in os.environ:
        sys.exit(self_launch(args))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if rank == 0:
            print('bench.py: --gpus %d but WORLD_SIZE=%d' % (args.gpus, world), file=sys.stderr)
        sys.exit(2)

    import numpy as np
    import torch
    import torch.distributed as dist
    from ksfd_amd import lib as klib
    from ksfd_amd.dist import open_handle

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
    zc = reference_rng().normal(size=tuple(max(1, n // 4) for n in cfg.n[:cfg.dim])) * 90.0

    def opts_for(adapt):
        o = klib.default_step_opts(adapt=adapt, atol=0.01, rtol=1e-6)        # options84:18-19
        if args.ksp_rtol > 0:
            o.ksp_rtol = args.ksp_rtol
        if args.pc_type >= 0:
            o.pc_type = args.pc_type
        return o

    spacing = cfg.spacing

    def one_step(t, h, opts):
        t, h, st, rc = ks.step(t, h, opts)                   # groom + ROSW/GMRES step (+ rejections)
        vmax = ks.velocity_max()                              # CFL_check, KSFD/ksfdts.py:287-319
        cfl = min(s * 2 / v if v > 0 else float('inf') for s, v in zip(spacing, vmax[:cfg.dim]))
        return t, h, st, cfl

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        ks.synchronize()

    def max_over_ranks(x):
        if world == 1:
            return x
        tt = torch.tensor([x], dtype=torch.float64, device='cuda' if args.dist_backend == 'nccl' else 'cpu')
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    class Tally:
        def __init__(self):
            self.steps = self.its = self.rej = self.launches = self.syncs = self.resid = self.pred = 0
            self.hs, self.pcs, self.bytes = [], {}, 0.0

        def add(self, st):
            self.steps += 1
            self.its += st.linear_its
            self.rej += st.rejections
            self.launches += st.launches
            self.syncs += st.host_syncs
            self.resid += st.residual_evals
            self.pred += st.predicted_final
            self.hs.append(st.h_used)
            self.bytes += st.bytes
            for bit, nm in PC_NAMES.items():
                if st.pc_used & bit:
                    self.pcs[nm] = self.pcs.get(nm, 0) + 1

        def summary(self):
            n = max(self.steps, 1)
            # linear_its: applications of the spectral inverse (sweeps) where the spectral solver ran, GMRES iterations elsewhere
            return {'steps': self.steps, 'linear_its_per_step': self.its / n, 'rejections': self.rej,
                    'h_min': float(np.min(self.hs)), 'h_max': float(np.max(self.hs)), 'h_mean': float(np.mean(self.hs)),
                    'steps_by_solver': self.pcs, 'residual_evals_per_step': self.resid / n, 'predicted_final_sweeps_per_step': self.pred / n,
                    'launches_per_step': self.launches / n, 'host_syncs_per_step': self.syncs / n}

    # ---------------- fixed-h leg: 100 steps at h = 1e-3 from the start values
    fixed = None
    if args.fixed_h > 0 and args.fixed_steps > 0:
        ks.set_state_random(zc, 9000.0)
        o = opts_for(0)
        t, h = 0.0, args.fixed_h
        for _ in range(3):
            t, h, st, cfl = one_step(t, args.fixed_h, o)
        barrier()
        t0 = time.perf_counter()
        tal = Tally()
        for _ in range(args.fixed_steps):
            t, h, st, cfl = one_step(t, args.fixed_h, o)
            tal.add(st)
        barrier()
        el = max_over_ranks(time.perf_counter() - t0)
        fixed = {'h': args.fixed_h, 'steps': args.fixed_steps, 'ms_per_step': 1e3 * el / args.fixed_steps,
                 'value': cfg.N * args.fixed_steps / el, 'linear_its_per_step': tal.its / args.fixed_steps,
                 'steps_by_solver': tal.pcs}

    # ---------------- ramp from dt0 to t* (setup of the pinned window; reported, not the headline)
    ks.set_state_random(zc, 9000.0)
    o = opts_for(1)
    t, h = 0.0, args.dt0
    barrier()
    t0 = time.perf_counter()
    tal = Tally()
    while t < args.t_star and tal.steps < 400:
        t, h, st, cfl = one_step(t, h, o)
        tal.add(st)
    barrier()
    el = max_over_ranks(time.perf_counter() - t0)
    ramp = dict(tal.summary(), dt0=args.dt0, t_end=t, seconds=el, value=cfg.N * tal.steps / el, ms_per_step=1e3 * el / tal.steps)
    t_star, h_star = t, h
    ks.checkpoint()

    # ---------------- pinned window: warm-up with events on every launch (per-class table), then the timed region with
    # events only around the dominant class (each event pair costs ~8 us of device time)
    def walk(nsteps, tally=None):
        t, h = t_star, h_star
        for i in range(nsteps):
            if i % args.window == 0:
                ks.restore()
                t, h = t_star, h_star
            t, h, st, cfl = one_step(t, h, o)
            if tally is not None:
                tally.add(st)
        return t

    # ---------------- N > 1: which stage solver for THIS interconnect?  The slab-distributed spectral solver moves the whole work array
    # through two all-to-alls per sweep (24 per step); polynomially preconditioned GMRES only exchanges ghost rows.  Which one is faster
    # depends on the links between the devices, so both walk the window once (after a warm-up walk each) and the faster one stays.
    # Same tolerance and stopping test either way; the line reports both times and what ran (steps_by_solver).
    autotune = None
    rehearse = bool(os.environ.get('KSFD_BENCH_AUTOTUNE'))     # rehearsal on a shared GPU (host transport: open_handle keeps the spectral solver off there)
    if rehearse and world > 1:
        ks.set_spectral_params(enable=1)
    if world > 1 and args.pc_type < 0 and not args.no_autotune and (getattr(ks, 'spectral_distributed', False) or rehearse):
        def timed_walk():
            walk(args.window)
            barrier()
            t0_ = time.perf_counter()
            walk(args.window)
            barrier()
            return max_over_ranks(time.perf_counter() - t0_) / args.window
        ms_spec = 1e3 * timed_walk()
        ks.set_spectral_params(enable=0)
        ms_alt = 1e3 * timed_walk()
        keep_spectral = ms_spec <= ms_alt                      # max over ranks: the same decision everywhere
        if keep_spectral:
            ks.set_spectral_params(enable=1)
        autotune = {'spectral_ms_per_step': ms_spec, 'without_spectral_ms_per_step': ms_alt, 'chosen': 'spectral' if keep_spectral else 'gmres (polynomial / multigrid)'}
        ks.restore()

    ks.set_profiling(True)
    ks.profile(reset=True)
    walk(args.warmup)
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
    tal = Tally()
    barrier()
    t_start = time.perf_counter()
    t_last = walk(args.steps, tal)
    barrier()
    elapsed = max_over_ranks(time.perf_counter() - t_start)
    prof = ks.profile()
    ks.set_profiling(False)

    aggregated = None
    if world == 1 and args.dim == 2 and args.agg_tile > 0 and args.agg_steps > 0:
        ks.close()                                            # its memory is not needed any more
        ks = None
        aggregated = aggregated_leg(args, klib, np)

    if rank == 0:
        N = cfg.N
        value = N * args.steps / elapsed
        if dom is None:
            dom = max((k for k in prof if k not in skip), key=lambda k: prof[k]['ms'])
        d = prof[dom]
        nl = max(d['launches'], 1)
        per_launch_ms = d['ms'] / nl
        impl_bytes, alg_bytes = d['bytes'] / nl, d['alg_bytes'] / nl
        gbs = lambda b: b / (per_launch_ms * 1e-3) / 1e9 if per_launch_ms > 0 else 0.0
        table = warm if args.warmup > 0 else prof             # all-class timings: warm-up steps (see above)
        tot_ms = sum(v['ms'] for v in table.values()) or 1.0
        kern = {k: dict(ms=round(v['ms'], 3), launches=int(v['launches']), share=round(v['ms'] / tot_ms, 3),
                        GBs=round(v['bytes'] / (v['ms'] * 1e-3) / 1e9, 1) if v['ms'] > 0 else None,
                        GBs_algorithmic=round(v['alg_bytes'] / (v['ms'] * 1e-3) / 1e9, 1) if v['ms'] > 0 else None)
                for k, v in table.items() if v['launches']}
        default_workload = world == 1 and args.n == 4096 and args.dim == 2 and args.nlig == 1
        traffic, traffic_src = pmc_traffic(dom) if default_workload else (None, None)
        live_err = None
        if default_workload and not args.no_live_pmc:
            lt, live_err = live_pmc(args, dom)
            if lt:
                traffic, traffic_src, live_err = lt, 'live: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE child runs of this command (ramp + 7 steps of the window), 2*FETCH_SIZE + WRITE_SIZE', None
        rp_us, rp_src = rocprof_avg_us(dom) if default_workload else (None, None)
        out = {
            'metric': 'grid-point-updates/sec (implicit step)', 'value': value, 'unit': 'grid-point-updates/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps,
            'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': dict({'workload': '%dD %s %d-ligand Keller-Segel (options84 spacing/physics), implicit step = 4-stage ROSW RA34PW2 '
                                        '(frozen Jacobian, TSAdaptBasic) with matrix-free stage solves on the device [%s] + CFL check; '
                                        'pinned window: the %d adaptive (rtol=1e-6, atol=0.01) steps that follow model time t*=%g of '
                                        'the run from dt0=%g, walked cyclically' % (
                                            args.dim, 'x'.join([str(args.n)] * args.dim), args.nlig,
                                            ' | '.join('%s: %d of %d steps' % (k, v, tal.steps) for k, v in sorted(tal.pcs.items())),
                                            args.window, args.t_star, args.dt0),
                             'stage_solver': {k: SOLVER_TEXT[k] for k in sorted(tal.pcs)},
                             'mixed_precision': MIXED_PRECISION,
                             'grid': [args.n] * args.dim, 'fields': cfg.F, 'ksp_rtol': float(o.ksp_rtol), 'pc_type': int(o.pc_type),
                             't_star': t_star, 'h_star': h_star, 'window': args.window, 't_end_of_window': t_last,
                             'parallelism': 'slab%d' % world, 'transport': getattr(ks, 'transport_name', 'none'),
                             'rccl_error': getattr(ks, 'rccl_error', None),
                             'spectral_distributed': getattr(ks, 'spectral_distributed', None), 'solver_autotune': autotune}, **tal.summary()),
            'roofline': {'bound': 'hbm', 'kernel': dom,
                         # `achieved`/`frac`: ALGORITHMIC bytes per launch (Jacobian action 24*F*N) / HIP-event time per launch
                         'achieved': gbs(alg_bytes), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': gbs(alg_bytes) / HBM_PEAK_GBS,
                         'frac_algorithmic': gbs(alg_bytes) / HBM_PEAK_GBS,
                         # bytes the implementation has to move per launch (frozen-coefficient planes, fused operands)
                         'frac_implementation': gbs(impl_bytes) / HBM_PEAK_GBS,
                         'traffic': traffic, 'traffic_source': traffic_src, 'traffic_live_error': live_err,
                         'alg_bytes_per_launch': alg_bytes, 'impl_bytes_per_launch': impl_bytes, 'ms_per_launch': per_launch_ms,
                         'launches': int(d['launches']), 'rocprof_avg_us': rp_us, 'rocprof_source': rp_src,
                         # whole step over the wall clock of the timed region: bytes the implementation moves, and SURVEY 8d's
                         # algorithmic bytes of the same launches (every class)
                         'step_implementation_GBs': tal.bytes / elapsed / 1e9,
                         'step_frac_of_peak': tal.bytes / elapsed / 1e9 / HBM_PEAK_GBS,
                         'step_algorithmic_GBs': sum(v['alg_bytes'] for v in prof.values()) / elapsed / 1e9,
                         'step_frac_algorithmic': sum(v['alg_bytes'] for v in prof.values()) / elapsed / 1e9 / HBM_PEAK_GBS},
            'ramp': ramp, 'fixed_h': fixed, 'aggregated': aggregated,
            'kernels': kern, 'kernels_region': 'warmup steps (events on every launch)' if args.warmup > 0 else 'timed steps',
        }
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(args, float(np.mean(tal.hs)))
            if args.cpu_lu_n > 0 and args.dim == 2:
                out['cpu_baseline_lu'] = cpu_baseline_lu(args, float(np.mean(tal.hs)))
        print(json.dumps(out), flush=True)
    if ks is not None:
        ks.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def aggregated_leg(args, klib, np):
    """Late phase of a production run on one GPU (where DESIGN.md 2 says the wall clock of a long run goes).  A (tile)^2 run with
    options81 spacing (width 1 at 384 points, options81:17) from the synthetic start values and dt0 = 1e-8 until aggregates have
    formed (rho spread over more than a decade and the stepper has left the spectral regime), then that state tiled periodically to
    (agg_n)^2 -- a periodic tiling is an exact solution of the larger problem -- and agg_steps adaptive steps timed there."""
    from ksfd_amd.config import ProblemConfig
    from ksfd_amd.initial import start_values
    m, n = args.agg_tile, args.agg_n
    reps = max(1, n // m)
    n = m * reps
    small = ProblemConfig.standard(2, (m, m), L=(m / 384.0,) * 2, nlig=args.nlig)
    k1 = klib.KSFDHip(small)
    k1.set_state(start_values(small))
    opts = klib.default_step_opts(adapt=1, atol=0.01, rtol=1e-6)
    t, h, nst = 0.0, 1e-8, 0
    t0 = time.perf_counter()
    mg_steps = 0
    while nst < 600 and mg_steps < 60:                        # 60 steps into the regime the fallback solver owns
        t, h, st, rc = k1.step(t, h, opts, raise_on_error=False)
        nst += 1
        if rc:
            break
        if st.pc_used & 2:
            mg_steps += 1
    setup_s = time.perf_counter() - t0
    u = k1.get_state().reshape(small.F, m, m)
    k1.close()
    rho = u[0]
    big = ProblemConfig.standard(2, (n, n), L=(n / 384.0,) * 2, nlig=args.nlig)
    kb = klib.KSFDHip(big)
    kb.set_state(np.tile(u, (1, reps, reps)).reshape(-1))
    for _ in range(2):
        t, h, st, rc = kb.step(t, h, opts)
    kb.synchronize()
    t0 = time.perf_counter()
    its, pcs, hs, rej = 0, {}, [], 0
    for _ in range(args.agg_steps):
        t, h, st, rc = kb.step(t, h, opts)
        kb.velocity_max()
        its += st.linear_its
        rej += st.rejections
        hs.append(st.h_used)
        for bit, nm in PC_NAMES.items():
            if st.pc_used & bit:
                pcs[nm] = pcs.get(nm, 0) + 1
    kb.synchronize()
    el = time.perf_counter() - t0
    kb.close()
    return {'grid': [n, n], 'fields': big.F, 'steps': args.agg_steps, 'ms_per_step': 1e3 * el / args.agg_steps,
            'value': big.N * args.agg_steps / el, 'unit': 'grid-point-updates/s', 'linear_its_per_step': its / args.agg_steps,
            'steps_by_solver': pcs, 'rejections': rej, 'h_mean': float(np.mean(hs)), 't': t,
            'rho_min': float(rho.min()), 'rho_max': float(rho.max()),
            'state': '%dx%d run (options81 spacing) from dt0=1e-8, %d steps in %.1f s to t=%.4g, tiled %dx%d' % (m, m, nst, setup_s, t, reps, reps)}


PMC_CLASS_PATTERNS = {'spectral': ('k_spec',), 'jvp': ('k_jvp',), 'rhs': ('k_rhs',), 'multidot': ('k_multidot',), 'mg': ('k_mg_', 'k_cheb', 'k_restrict', 'k_prolong')}


def live_pmc(args, kernel_class):
    """HBM bytes per launch of a kernel class measured in THIS invocation: two child runs of this very command (same grid and window, 5 timed
    steps, no side legs) under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` -- separate passes, counters only beside the kernel trace,
    as /opt/skills/guides/MI355X_MICROARCH.md prescribes; gfx950: FETCH_SIZE counts the 128-B requests of 16-B/lane reads at 64 B, hence
    2*FETCH_SIZE + WRITE_SIZE (KB).  Returns (bytes per launch or None, error text or None); never raises."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    pats = PMC_CLASS_PATTERNS.get(kernel_class)
    if not pats:
        return None, 'no kernel-name pattern for class %s' % kernel_class
    if any('rocprof' in os.environ.get(k, '') for k in ('LD_PRELOAD', 'ROCP_TOOL_LIBRARIES')):
        return None, 'already running under a profiler'
    exe = shutil.which('rocprofv3') or ('/opt/rocm/bin/rocprofv3' if os.path.exists('/opt/rocm/bin/rocprofv3') else None)
    if not exe:
        return None, 'rocprofv3 not found'
    avg = {}
    for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
        d = tempfile.mkdtemp(prefix='ksfd_pmc_', dir='/tmp')
        try:
            cmd = [exe, '--pmc', counter, '--kernel-trace', '--output-format', 'csv', '-d', d, '--', sys.executable, os.path.abspath(__file__),
                   '--gpus', '1', '--steps', '5', '--warmup', '2', '--grid', str(args.n), '--nlig', str(args.nlig), '--dim', str(args.dim),
                   '--no-cpu-baseline', '--no-live-pmc', '--agg-tile', '0', '--fixed-h', '0']
            r = subprocess.run(cmd, cwd='/tmp', env=dict(os.environ, TMPDIR='/tmp'), timeout=240, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
            if r.returncode != 0:
                return None, 'rocprofv3 --pmc %s exited with %d: %s' % (counter, r.returncode, r.stderr.decode(errors='replace')[-200:])
            n, tot = 0, 0.0
            for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
                with open(f) as fh:
                    for row in csv.DictReader(fh):
                        if row.get('Counter_Name') == counter and any(p_ in row['Kernel_Name'] for p_ in pats):
                            n += 1
                            tot += float(row['Counter_Value'])
            if n == 0:
                return None, 'no %s rows for class %s' % (counter, kernel_class)
            avg[counter] = tot / n
        except Exception as e:                                   # a profiler problem must never cost the bench line
            return None, '%s: %s' % (type(e).__name__, e)
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return (2.0 * avg['FETCH_SIZE'] + avg['WRITE_SIZE']) * 1024.0, None


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


def rocprof_avg_us(kernel_class):
    """Average duration of this class' kernels in the newest committed rocprofv3 --kernel-trace --stats summary of this
    command (profiles/*_pmc.json carries it as 'avg_us' per class; tools/summarize_prof.py)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_pmc.json')))
    if not files:
        return None, None
    try:
        c = json.load(open(files[-1]))['classes'].get(kernel_class)
        return (c.get('avg_us'), os.path.basename(files[-1])) if c else (None, None)
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
    """The oracle's restatement of the same step (kind 'port': ROSW + unpreconditioned GMRES(30), classic CGS2), OpenMP over
    the host cores this process may use, on a bounded sample: ONE step at the pinned window's mean h on an m x m sub-grid
    with the same spacing, physics and initial-data statistics.  The iteration count of the unpreconditioned solve grows
    with h, so the iteration cap bounds the leg (a step that hits the cap is reported as such, with its partial rate)."""
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
    rt = args.ksp_rtol if args.ksp_rtol > 0 else 1e-6
    t0 = time.perf_counter()
    note = ''
    try:
        un, err, wr, its = o.rosw_step(u, h, 0.01, 1e-6, solver='gmres', ksp_rtol=rt, restart=30, maxit=1200)
        note = '%d GMRES its' % its
    except RuntimeError:
        note = 'iteration cap (1200) reached: rate is an upper bound'
    dt = time.perf_counter() - t0
    ko.set_threads(1)
    return {'value': cfg.N / dt, 'unit': 'grid-point-updates/s', 'cores': cores, 'kind': 'port',
            'sample': 'one ROSW+GMRES(30,CGS2, no preconditioner) step at h=%.4g on a %s sub-grid (same spacing/physics/IC '
                      'statistics), %s, %.1f s' % (h, 'x'.join([str(m)] * args.dim), note, dt)}




def cpu_baseline_lu(args, h):
    """The REFERENCE's algorithm for one step, on the CPU: Jacobian assembled entry by entry (the oracle's restatement of
    Derivatives.Jacobian + ksfdMat.setValuesJacobian), shift*I - J factorised ONCE by a sparse direct solver, four back-substitutions
    (-ksp_type preonly -pc_type lu, options84:58-60; PETSc hands the factorisation to MUMPS, here scipy's SuperLU with minimum-degree
    ordering, one thread).  Bounded sample: SuperLU needs ~15 s for 256^2 x 2 unknowns and > 80 s for 512^2, so 256^2 it is."""
    import numpy as np
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    from ksfd_amd.initial import start_values
    from oracle import ko
    m = args.cpu_lu_n
    cfg = build_problem(m, args.nlig, dim=2)
    u = start_values(cfg)
    o = ko.Oracle(cfg)
    At, Gi, bt, b2t, asum = ko.tableau()
    gam = 1.0 / Gi[0, 0]
    F, N = cfg.F, cfg.N
    to_vec = lambda a: a.reshape(F, N).T.reshape(-1)
    to_soa = lambda x: x.reshape(N, F).T.reshape(-1)
    t0 = time.perf_counter()
    ug = o.groom(u)
    rp, col, val = o.jacobian_csr(ug)
    J = sp.csr_matrix((val, col, rp), shape=(F * N, F * N))
    lu = spla.splu((sp.identity(F * N, format='csc') / (gam * h) - J).tocsc(), permc_spec='MMD_AT_PLUS_A')
    t_fact = time.perf_counter() - t0
    Y = []
    for i in range(4):
        Z = ug + sum(At[i, j] * Y[j] for j in range(i))
        Zdot = sum((Gi[i, j] / h) * Y[j] for j in range(i)) if i else 0.0
        Y.append(to_soa(lu.solve(to_vec(o.rhs(Z) - Zdot))))
    dt = time.perf_counter() - t0
    return {'value': cfg.N / dt, 'unit': 'grid-point-updates/s', 'cores': 1, 'kind': 'port',
            'sample': 'one ROSW step at h=%.4g on a %dx%d sub-grid (same spacing/physics/IC statistics): assembled Jacobian + one sparse LU '
                      '(SuperLU, MMD ordering, %.1f s, fill %.3g nonzeros) + four back-substitutions, %.1f s in all'
                      % (h, m, m, t_fact, lu.L.nnz + lu.U.nnz, dt)}


if __name__ == '__main__':
    main()
