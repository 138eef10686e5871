import sys
sys.path.insert(0, '.')
import numpy as np
from ksfd_amd import lib as klib
from ksfd_amd.config import ProblemConfig
sp = 0.08 / 32
for n in (32, 128, 512):
    for nlig in (1, 2):
        for amp in (90.0, 900.0):
            cfg = ProblemConfig.standard(2, (n, n), L=(n * sp, n * sp), nlig=nlig)
            rng = np.random.default_rng(4); N = n * n
            rho = 9000 + amp * rng.standard_normal(N)
            u = np.concatenate([rho] + [rho * cfg.lig_s[l] / cfg.lig_gamma[l] for l in range(nlig)])
            k = klib.KSFDHip(cfg)
            out = []
            for h in (0.5, 50.0):
                k.set_state(u)
                t, hn, st, rc = k.step(0.0, h, klib.default_step_opts(adapt=0, atol=0.01, rtol=1e-6, ksp_rtol=1e-8, pc_type=1), raise_on_error=False)
                out.append('h=%g: rc %d its %d' % (h, rc, st.linear_its))
            print('n %4d nlig %d amp %4g  ' % (n, nlig, amp), ' | '.join(out), flush=True)
            k.close()
