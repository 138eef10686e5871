"""Stiff-ish fixed-h steps: multigrid vs Chebyshev polynomial preconditioner (wall time per step)."""
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
import os
if os.environ.get('KSFD_TUNE'):
    ks.set_tuning(use_fused=int(os.environ['KSFD_TUNE']))
if os.environ.get('KSFD_MG'):                      # nu,ncoarse_max,ratio,coarse_tol
    a = os.environ['KSFD_MG'].split(',')
    ks.set_mg_params(nu=int(a[0]), ncoarse_max=int(a[1]), ratio=float(a[2]), coarse_tol=float(a[3]))
for h in (0.02, 0.05, 0.1, 0.2, 0.5, 1.0):
    for name, pc, deg, tgt in (('mg', 1, 3, 0.02), ('poly3', 3, 3, 0.02), ('poly6', 3, 6, 0.02), ('poly6t', 3, 6, 0.1), ('poly4t', 3, 4, 0.1)):
        ks.set_poly_params(deg, tgt)
        best, its = None, 0
        for rep in range(2):
            ks.set_state(u0)
            o = klib.default_step_opts(adapt=0, atol=0.01, rtol=1e-6, ksp_rtol=1e-6, pc_type=pc, ksp_max_it=3000)
            ks.synchronize(); t0 = time.perf_counter()
            t, hn, st, rc = ks.step(0.0, h, o, raise_on_error=False)
            ks.synchronize(); dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        print('n %d h %-5g X %-6.1f %-6s rc %d outer %4d jvp %4d  %.1f ms' % (n, h, 190.5 * h * (4096 / n) ** 0, name, rc, st.linear_its, st.jvp_evals, best * 1e3), flush=True)
