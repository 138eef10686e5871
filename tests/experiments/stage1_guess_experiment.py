"""GPU experiment (not a test): how well is the FIRST stage system of a step, (shift_n I - J_n) Y = f(u_n), predicted from the first stages of
the previous steps?  b~_j = b_j + (shift_n - shift_j) Y_j is what A_n Y_j would be but for the change of J; c = argmin ||b_n - sum c_j b~_j||,
x0 = sum c_j Y_j.  Prints the relative size of the assumed residual b_n - sum c_j b~_j and of the true one b_n - A_n x0, for 1..3 previous steps,
in the window the bench measures."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
from bench import build_problem
from ksfd_amd import lib as klib
from ksfd_amd.initial import reference_rng

GAMMA = 4.3586652150845900e-01
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
cfg = build_problem(n, 1)
zc = reference_rng().normal(size=(256, 256)) * 90.0
k = klib.KSFDHip(cfg)
k.set_state_random(zc, 9000.0)
o = klib.default_step_opts(adapt=1, atol=0.01, rtol=1e-6, ksp_rtol=1e-10)
t, h = 0.0, 1e-8
hist = []
nrm = np.linalg.norm
while len(hist) < 8:
    if t >= 0.35:
        u = k.get_state()
        s = 1.0 / (GAMMA * h)
        b = k.rhs(u)
        # Y = A^-1 b by defect correction with the library's own operators (frozen J at u)
        x = np.zeros_like(b); r = b.copy()
        for it in range(12):
            x += k.spectral_apply(s, r)
            r = b - (s * x - k.jvp(x, u))
            if nrm(r) < 1e-11 * nrm(b):
                break
        hist.append((t, h, s, b, x))
        if len(hist) > 1:
            line = 't %.4f h %.4f (h/hprev %.3f):' % (t, h, h / hist[-2][1])
            for m in (1, 2, 3):
                if len(hist) > m:
                    prev = hist[-1 - m:-1]
                    Bt = np.stack([pb + (s - ps) * px for (_, _, ps, pb, px) in prev], 1)
                    c, *_ = np.linalg.lstsq(Bt, b, rcond=None)
                    x0 = sum(ci * p[4] for ci, p in zip(c, prev))
                    ra = b - Bt @ c
                    rt = b - (s * x0 - k.jvp(x0, u))
                    line += '  m=%d assumed %.2e true %.2e c=%s' % (m, nrm(ra) / nrm(b), nrm(rt) / nrm(b), np.round(c, 3))
            print(line, flush=True)
            # what a zero start gives after one sweep, for comparison
            x1 = k.spectral_apply(s, b)
            print('     zero start: after sweep 0 %.2e' % (nrm(b - (s * x1 - k.jvp(x1, u))) / nrm(b)), flush=True)
    # the step itself is taken from the state the handle holds
    t2, h2, st, rc = k.step(t, h, o)
    t, h = t2, h2
k.close()
