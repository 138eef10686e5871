"""Late (aggregated) phase on one GPU, by kernel class: the state bench.py's aggregated leg builds (tile^2 run with options81 spacing until
the V cycle has owned 60 steps, tiled to n^2), then `steps` adaptive steps with events on every launch.
usage: python tools/agg_profile.py [n=2048] [nlig=1] [steps=5] [tile=256]"""
import sys, os, time, json
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from ksfd_amd import lib as klib
from ksfd_amd.config import ProblemConfig
from ksfd_amd.initial import start_values

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
nlig = int(sys.argv[2]) if len(sys.argv) > 2 else 1
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
m = int(sys.argv[4]) if len(sys.argv) > 4 else 256
reps = n // m
state_file = os.environ.get('AGG_STATE')
if state_file and os.path.exists(state_file):
    z = np.load(state_file)
    u, t, h = z['u'], float(z['t']), float(z['h'])
else:
    small = ProblemConfig.standard(2, (m, m), L=(m / 384.0,) * 2, nlig=nlig)
    k1 = klib.KSFDHip(small)
    k1.set_state(start_values(small))
    opts = klib.default_step_opts(adapt=1, atol=0.01, rtol=1e-6)
    t, h, nst, mg = 0.0, 1e-8, 0, 0
    while nst < 600 and mg < 60:
        t, h, st, rc = k1.step(t, h, opts, raise_on_error=False)
        nst += 1
        if rc:
            break
        mg += 1 if st.pc_used & 2 else 0
    u = k1.get_state().reshape(small.F, m, m)
    k1.close()
    if state_file:
        np.savez(state_file, u=u, t=t, h=h)
        print('saved', state_file)
        sys.exit(0)
opts = klib.default_step_opts(adapt=1, atol=0.01, rtol=1e-6)
print('tile state: t=%.4g h=%.4g rho %.0f..%.0f' % (t, h, u[0].min(), u[0].max()), flush=True)
big = ProblemConfig.standard(2, (n, n), L=(n / 384.0,) * 2, nlig=nlig)
kb = klib.KSFDHip(big)
kb.set_state(np.tile(u, (1, reps, reps)).reshape(-1))
if os.environ.get('MG_NU'):
    kb.set_mg_params(nu=int(os.environ['MG_NU']))
if os.environ.get('MG_RATIO'):
    kb.set_mg_params(ratio=float(os.environ['MG_RATIO']))
if os.environ.get('KSFD_TUNE_BITS'):
    kb.set_tuning(use_fused=1 | int(os.environ['KSFD_TUNE_BITS']))
if os.environ.get('MG_EAGER'):
    kb.set_mg_params(power_its=-7)          # no hipGraph for the coarse levels (rocprofv3 cannot trace the capture)
for _ in range(2):
    t, h, st, rc = kb.step(t, h, opts)
kb.synchronize()
t0 = time.perf_counter()
its = 0; launches = 0; syncs = 0
for _ in range(steps):
    t, h, st, rc = kb.step(t, h, opts)
    its += st.linear_its; launches += st.launches; syncs += st.host_syncs
kb.synchronize()
el = time.perf_counter() - t0
print('%d^2 x %d fields: %.2f ms/step, %.1f its/step (%.3f ms per iteration), %.0f launches and %.1f host syncs per step, pc_used %d, h %.3g'
      % (n, big.F, 1e3 * el / steps, its / steps, 1e3 * el / max(its, 1), launches / steps, syncs / steps, st.pc_used, st.h_used), flush=True)
if os.environ.get('AGG_CLASSES', '1') != '0':
    kb.set_profiling(True)
    for _ in range(steps):
        t, h, st, rc = kb.step(t, h, opts)
    prof = kb.profile()
    tot = sum(v['ms'] for v in prof.values())
    for name, v in sorted(prof.items(), key=lambda kv: -kv[1]['ms']):
        print('  %-14s %8.2f ms/step  %6d launches/step  %5.1f %%  %7.0f GB/s' % (name, v['ms'] / steps, v['launches'] // steps, 100 * v['ms'] / tot, v.get('GBs', 0.0)))
    print('  sum of classes %.2f ms/step' % (tot / steps))
kb.close()
