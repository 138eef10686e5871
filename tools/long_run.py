"""Production-style run: adaptive integration from dt=1e-8 far into the aggregation phase (options81-style)."""
import sys, time
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from ksfd_amd import lib as klib
from ksfd_amd.config import ProblemConfig
from ksfd_amd.initial import start_values
n = int(sys.argv[1]) if len(sys.argv) > 1 else 384
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
nlig = int(sys.argv[3]) if len(sys.argv) > 3 else 2
cfg = ProblemConfig.standard(2, (n, n), L=(n / 384.0, n / 384.0), nlig=nlig)      # options81 spacing (width 1 at 384)
ks = klib.KSFDHip(cfg)
ks.set_state(start_values(cfg))
if len(sys.argv) > 4:
    ks.set_mg_params(nu=int(sys.argv[4]))
import os
if os.environ.get('KSFD_TUNE'):
    ks.set_tuning(use_fused=int(os.environ['KSFD_TUNE']))
if os.environ.get('KSFD_NO_SPECTRAL'):
    ks.set_spectral_params(enable=0)
opts = klib.default_step_opts(adapt=1, atol=0.01, rtol=1e-6)
t, h = 0.0, 1e-8
T0 = time.perf_counter()
tot_its = 0
prof_left = None
for s in range(nsteps):
    t0 = time.perf_counter()
    t, h, st, rc = ks.step(t, h, opts, raise_on_error=False)
    tot_its += st.linear_its
    if rc:
        print('STOP rc', rc, ks.last_error()); break
    if s % 10 == 0 or st.rejections:
        vm = ks.velocity_max()
        print('step %4d t %.4e h_next %.3e its %4d rej %d wrms %.2e  %.1f ms  vmax %.2e pc %d' % (s, t, h, st.linear_its, st.rejections, st.wrms, 1e3 * (time.perf_counter() - t0), vm[0], st.pc_used), flush=True)
    if t > 2e5: break
    # KSFD_PROFILE_MG=k: per-kernel-class times of k steps once the multigrid regime has begun
    if os.environ.get('KSFD_PROFILE_MG') and (st.pc_used & 2) and prof_left is None:
        prof_left = int(os.environ['KSFD_PROFILE_MG']); ks.set_profiling(True); ks.profile(reset=True); prof_its0 = tot_its; prof_s0 = s
    elif prof_left is not None and prof_left > 0:
        prof_left -= 1
        if prof_left == 0:
            pr = ks.profile(reset=True); ks.set_profiling(False)
            tot = sum(v['ms'] for v in pr.values())
            print('--- kernel classes over steps %d..%d (%d its): %.1f ms of events' % (prof_s0 + 1, s, tot_its - prof_its0, tot))
            for k, v in sorted(pr.items(), key=lambda kv: -kv[1]['ms']):
                if v['launches']: print('    %-12s %9.2f ms %7d launches  %6.1f GB/s' % (k, v['ms'], v['launches'], v['bytes'] / max(v['ms'], 1e-9) / 1e6), flush=True)
u = ks.get_state()
print('done: steps %d t %.4e total its %d wall %.1f s  rho min %.3g max %.3g' % (s + 1, t, tot_its, time.perf_counter() - T0, u[:n*n].min(), u[:n*n].max()))
