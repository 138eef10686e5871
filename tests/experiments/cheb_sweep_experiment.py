"""GPU experiment (not a test): residual histories of the defect correction x += M^-1 r for the first stage system in the bench window,
plain vs Chebyshev-accelerated for a spectrum of I - A M^-1 on the imaginary segment [-i rho, i rho] (Manteuffel's recurrence with
d = 1, c^2 = -rho^2), and vs the minimal-residual combination of the plain iterates (what GMRES would give at best)."""
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
o = klib.default_step_opts(adapt=1, atol=0.01, rtol=1e-6)
t, h = 0.0, 1e-8
nrm = np.linalg.norm
seen = 0
while seen < 3:
    if t >= 0.35:
        seen += 1
        u = k.get_state()
        s = 1.0 / (GAMMA * h)
        b = k.rhs(u)
        A = lambda v: s * v - k.jvp(v, u)
        Mi = lambda v: k.spectral_apply(s, v)
        bn = nrm(b)
        # plain
        x = np.zeros_like(b); r = b.copy(); hist = []; R = [b.copy()]
        for it in range(4):
            x = x + Mi(r); r = b - A(x); hist.append(nrm(r) / bn); R.append(r.copy())
        print('t %.3f h %.3f plain     %s' % (t, h, ' '.join('%.2e' % v for v in hist)))
        ratios = [hist[i + 1] / hist[i] for i in range(1, 3)]
        # minimal residual over the affine span of the plain iterates' residuals (GMRES-optimal within that span)
        mr = []
        for kk in range(1, 5):
            D = np.stack([R[j] - R[kk] for j in range(kk)], 1)
            c, *_ = np.linalg.lstsq(D, -R[kk], rcond=None)
            mr.append(nrm(R[kk] + D @ c) / bn)
        print('              min-res   %s' % ' '.join('%.2e' % v for v in mr))
        for rho in (max(ratios), 1.3 * max(ratios), 2.0 * max(ratios)):
            x = np.zeros_like(b); r = b.copy(); hist = []
            d = 1.0; c2 = -rho * rho
            delta = None; alpha = None
            for it in range(4):
                z = Mi(r)
                if it == 0:
                    alpha = 1.0 / d; delta = alpha * z
                else:
                    alpha = 2 * d / (2 * d * d - c2) if it == 1 else 1.0 / (d - 0.25 * c2 * alpha)
                    beta = d * alpha - 1.0
                    delta = alpha * z + beta * delta
                x = x + delta; r = b - A(x); hist.append(nrm(r) / bn)
            print('              cheb %.4f %s' % (rho, ' '.join('%.2e' % v for v in hist)))
    t, h, st, rc = k.step(t, h, o)
k.close()
