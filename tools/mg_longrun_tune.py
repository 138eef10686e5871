"""Multigrid knobs in the stiff small-grid regime (384^2 x 3 fields, h ~ 30): ms per step for steps 300..360."""
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
u300, t300, h300 = ks.get_state(), t, h
# every knob explicit in every row (ksfd_set_mg_params keeps a value when given <= 0); defaults: nu 2, ncoarse_max 400, ratio 6, coarse_tol 1e-2
for nu, ncm, ratio, ctol in [(2, 400, 6.0, 1e-2), (3, 400, 6.0, 1e-2), (2, 400, 10.0, 1e-2), (3, 400, 10.0, 1e-2), (2, 8, 6.0, 1e-2), (2, 16, 6.0, 1e-2),
                             (2, 32, 6.0, 1e-2), (2, 400, 6.0, 1e-1), (2, 400, 6.0, 3e-1), (3, 16, 10.0, 1e-1), (3, 8, 6.0, 1e-1), (2, 400, 6.0, 1e-2)]:
    ks.set_mg_params(nu=nu, ncoarse_max=ncm, ratio=ratio, coarse_tol=ctol)
    ks.set_state(u300); t, h = t300, h300
    ks.synchronize(); T0 = time.perf_counter(); its = 0
    for s in range(60):
        t, h, st, rc = ks.step(t, h, opts, raise_on_error=False); its += st.linear_its
        if rc: break
    ks.synchronize(); wall = time.perf_counter() - T0
    print('nu %d ncoarse_max %3d ratio %5.1f coarse_tol %g : rc %d %.2f ms/step %.1f its/step %.3f ms/it' % (nu, ncm, ratio, ctol, rc, 1e3 * wall / 60, its / 60, 1e3 * wall / max(its, 1)), flush=True)
