"""Where the time goes in the stiff, small-grid regime: 384^2 x 3 fields, steps 300..400 of the production-style run."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from ksfd_amd import lib as klib
from ksfd_amd.config import ProblemConfig
from ksfd_amd.initial import start_values
n = 384
cfg = ProblemConfig.standard(2, (n, n), L=(1.0, 1.0), nlig=2)
ks = klib.KSFDHip(cfg)
ks.set_state(start_values(cfg))
opts = klib.default_step_opts(adapt=1, atol=0.01, rtol=1e-6)
t, h = 0.0, 1e-8
for s in range(300):
    t, h, st, rc = ks.step(t, h, opts)
ks.synchronize(); T0 = time.perf_counter(); its = 0
for s in range(100):
    t, h, st, rc = ks.step(t, h, opts); its += st.linear_its
ks.synchronize(); wall = time.perf_counter() - T0
print('no profiling: %.2f ms/step, %.1f its/step, %.3f ms/it, h %.3g' % (10 * wall, its / 100, 1e3 * wall / its, h))
ks.set_profiling(True); ks.profile(reset=True)
T0 = time.perf_counter(); its = 0
for s in range(50):
    t, h, st, rc = ks.step(t, h, opts); its += st.linear_its
ks.synchronize(); wall = time.perf_counter() - T0
p = ks.profile()
tot = sum(v['ms'] for v in p.values())
print('with events: %.2f ms/step; kernel time %.2f ms/step; launches/step %.0f; its/step %.1f' % (20 * wall, tot / 50, sum(v['launches'] for v in p.values()) / 50, its / 50))
for k, v in sorted(p.items(), key=lambda kv: -kv[1]['ms']):
    if v['launches']:
        print('  %-12s %8.2f ms/step %7.1f launches/step  %6.1f us/launch' % (k, v['ms'] / 50, v['launches'] / 50, 1e3 * v['ms'] / v['launches']))
