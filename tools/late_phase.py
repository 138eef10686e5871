"""Solver experiments on the late, slowly converging phase of the 384^2 run: 30 adaptive steps from the saved state."""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np
from ksfd_amd import lib as klib
from ksfd_amd.config import ProblemConfig
z = np.load(sys.argv[1] if len(sys.argv) > 1 else 'tools/_scratch/late_state.npz')
nst = int(sys.argv[2]) if len(sys.argv) > 2 else 30
cfg = ProblemConfig.standard(2, (384, 384), L=(1.0, 1.0), nlig=2)
ks = klib.KSFDHip(cfg)
ks.set_state(z['u'])
if os.environ.get('KSFD_TUNE'):
    ks.set_tuning(use_fused=int(os.environ['KSFD_TUNE']))
if os.environ.get('KSFD_MG_NU'):
    ks.set_mg_params(nu=int(os.environ['KSFD_MG_NU']), ratio=float(os.environ.get('KSFD_MG_RATIO', '0')))
if os.environ.get('KSFD_MG_POWER'):
    ks.set_mg_params(power_its=int(os.environ['KSFD_MG_POWER']))
import os
opts = klib.default_step_opts(adapt=1, atol=0.01, rtol=1e-6, ksp_restart=int(os.environ.get('KSFD_RESTART', '30')))
t, h = float(z['t']), float(z['h'])
ks.synchronize(); T0 = time.perf_counter(); its = 0; rej = 0
for s in range(nst):
    t, h, st, rc = ks.step(t, h, opts, raise_on_error=False)
    its += st.linear_its; rej += st.rejections
    if rc: print('rc', rc, ks.last_error()); break
ks.synchronize(); wall = time.perf_counter() - T0
print('%d steps: t %.5g h %.4g  %.1f its/step  %.1f ms/step  rejections %d' % (s + 1, t, h, its / (s + 1), 1e3 * wall / (s + 1), rej))
