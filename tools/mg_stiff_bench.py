"""Stiff fixed-h steps: unpreconditioned GMRES vs multigrid-preconditioned (iterations, time, agreement)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from bench import build_problem
from ksfd_amd import lib as klib
from ksfd_amd.initial import start_values
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
hs = [float(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else [0.01, 0.1, 1.0, 10.0]
cfg = build_problem(n, 1)
u0 = start_values(cfg)
ks = klib.KSFDHip(cfg)
ks.set_profiling(True)
res = {}
for h in hs:
    for pc in (1, 0):
        ks.set_state(u0)
        o = klib.default_step_opts(adapt=0, atol=0.01, rtol=1e-6, ksp_rtol=1e-8, ksp_max_it=600, pc_type=pc)
        ks.synchronize(); t0 = time.perf_counter()
        t, hn, st, rc = ks.step(0.0, h, o, raise_on_error=False)
        ks.synchronize(); dt = time.perf_counter() - t0
        u = ks.get_state()
        res[(h, pc)] = u
        msg = ''
        if pc == 0 and (h, 1) in res and rc == 0:
            msg = 'relL2(mg vs none) %.2e' % (np.linalg.norm(res[(h, 1)] - u) / np.linalg.norm(u))
        pr = ks.profile(reset=True)
        print('   ', {k: (round(v['ms'], 1), v['launches']) for k, v in pr.items() if v['launches']})
        print('n %d h %-6g pc %d rc %d its %4d resid %.1e wrms %.3e time %.1f ms %s %s' %
              (n, h, pc, rc, st.linear_its, st.ksp_resid, st.wrms, dt * 1e3, msg, ks.last_error()[:80] if rc else ''), flush=True)
p = ks.profile()
print({k: (round(v['ms'], 1), v['launches']) for k, v in p.items() if v['launches']})
