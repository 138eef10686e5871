"""GPU experiment (not a test): the random step cases of tests/test_gpu_random_sweep.py WITHOUT the cap that keeps shift = 1/(gamma h)
above twice the growth rate of the chemotactic instability -- h is pushed so that shift falls BELOW the largest growth rates (stage
matrix indefinite), kept 2 % away from every eigenvalue of J so that the exact solve is well defined.  Which cases does the iterative
solver still get right, at which cost?"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np
import scipy.sparse as sp
from test_gpu_random_sweep import random_problem
from conftest import rel_l2
from ksfd_amd import lib as klib
from oracle import ko

GAM = 0.43586652150845900
fac = float(sys.argv[1]) if len(sys.argv) > 1 else 0.5          # shift = fac * growth  (0.5: well inside the indefinite range)
ok = bad = skipped = 0
for i in range(40):
    cfg, u, rng = random_problem(100 + i, for_step=True)
    rp, col, val = ko.Oracle(cfg).jacobian_csr(u)
    lam = np.linalg.eigvals(sp.csr_matrix((val, col, rp)).toarray())
    growth = float(lam.real.max())
    if growth <= 0.0:
        skipped += 1
        continue
    shift = fac * growth
    for _ in range(50):                                           # 2 % clear of every eigenvalue
        if np.abs(lam - shift).min() >= 0.02 * shift:
            break
        shift *= 1.03
    h = 1.0 / (GAM * shift)
    un, err, wr, _ = ko.Oracle(cfg).rosw_step(u, h, 0.01, 1e-6, solver='lu')
    k = klib.KSFDHip(cfg)
    k.set_state(u)
    t, hn, st, rc = k.step(0.0, h, klib.default_step_opts(adapt=0, atol=0.01, rtol=1e-6, ksp_rtol=1e-12, ksp_max_it=20000), raise_on_error=False)
    e = rel_l2(k.get_state(), un) if rc == 0 else float('nan')
    npos = int((lam.real > shift).sum())
    print('case %3d n %-12s F %d h %.3g shift/growth %.2f unstable modes above shift %3d: rc %d its %5d err %.2e %s' % (
        100 + i, cfg.n[:cfg.dim], cfg.F, h, shift / growth, npos, rc, st.linear_its, e, '' if (rc == 0 and e < 1e-9) else ' <--'), flush=True)
    ok += rc == 0 and e < 1e-9
    bad += not (rc == 0 and e < 1e-9)
    k.close()
print('shift = %.2f x growth: %d ok, %d not, %d without growth' % (fac, ok, bad, skipped))
