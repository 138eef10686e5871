"""Wall time of one stiff fixed-h step at 4096^2 vs multigrid smoothing parameters."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from bench import build_problem
from ksfd_amd import lib as klib
from ksfd_amd.initial import start_values
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cfg = build_problem(n, 1)
u0 = start_values(cfg)
ks = klib.KSFDHip(cfg)
for h in (0.3, 3.0):
    for nu, ratio, pw in ((2, 6.0, 8), (1, 6.0, 8), (3, 6.0, 8), (2, 4.0, 8), (2, 10.0, 8), (1, 4.0, 8), (2, 6.0, 4), (3, 10.0, 4)):
        ks.set_mg_params(nu=nu, ratio=ratio, power_its=pw)
        best = None
        for rep in range(2):
            ks.set_state(u0)
            o = klib.default_step_opts(adapt=0, atol=0.01, rtol=1e-6, ksp_rtol=1e-6, pc_type=1)
            ks.synchronize(); t0 = time.perf_counter()
            t, hn, st, rc = ks.step(0.0, h, o, raise_on_error=False)
            ks.synchronize(); dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        print('n %d h %-4g nu %d ratio %-4g power %d -> rc %d its %3d  %.1f ms' % (n, h, nu, ratio, pw, rc, st.linear_its, best * 1e3), flush=True)
