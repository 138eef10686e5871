"""How loose may the GMRES tolerance be?  One adaptive-size step from the bench state with ksp_rtol=1e-12
(stand-in for the reference's exact LU) vs looser tolerances: rel-L2 of the resulting fields."""
import sys
sys.path.insert(0, '.')
import numpy as np
from bench import build_problem
from ksfd_amd import lib as klib
from ksfd_amd.initial import start_values
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
cfg = build_problem(n, 1)
u0 = start_values(cfg)
ks = klib.KSFDHip(cfg)
def run(rt, nsteps=6, h0=0.05):
    ks.set_state(u0)
    o = klib.default_step_opts(adapt=1, atol=0.01, rtol=1e-6, ksp_rtol=rt)
    t, h, its, hs = 0.0, h0, 0, []
    for _ in range(nsteps):
        t, h, st, rc = ks.step(t, h, o)
        its += st.linear_its; hs.append(st.h_used)
    return ks.get_state(), its, t, hs
ref, its0, t0, hs0 = run(1e-12)
print('ref its', its0, 't', t0, 'hs', hs0)
for rt in (1e-10, 1e-8, 1e-7, 1e-6, 1e-5, 1e-4, 1e-3):
    u, its, t, hs = run(rt)
    print('ksp_rtol %.0e its %4d  t %.6f  relL2 vs tight %.3e  max|du| %.3e  (steps identical: %s)' %
          (rt, its, t, np.linalg.norm(u - ref) / np.linalg.norm(ref), np.abs(u - ref).max(), np.allclose(hs, hs0, rtol=1e-6)))
