"""CPU experiment (not a test): defect-correction histories on the near-uniform bench state with the plain constant-coefficient
inverse (what the GPU runs) against the SCALED one (unknowns (dG, v_U), rho row divided by rho: the highest-order term becomes
exactly constant-coefficient; tests/experiments/scaled_spectral_experiment.py).  Question: does a sweep contract enough more
to drop one sweep per stage solve?"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
from ksfd_amd.config import ProblemConfig
from ksfd_amd.initial import start_values
from oracle import ko
from fft_pc_experiment import symbol_D2
GAMMA = 4.3586652150845900e-01


def run(n, h, nlig=1, nsweep=5, amp=None):
    L = n * 4.0 / 1536
    cfg = ProblemConfig.standard(2, (n, n), L=(L, L), nlig=nlig)
    o = ko.Oracle(cfg)
    u = start_values(cfg)
    F, N = cfg.F, cfg.N
    if amp:
        u = u.reshape(F, -1).copy(); u[0] *= np.exp(amp * np.random.default_rng(1).standard_normal(N)); u = u.reshape(-1)
    ug = o.groom(u).reshape(F, n, n); rho = ug[0]
    s2 = cfg.s2; ms = cfg.maxscale * s2
    th = np.tanh((rho - cfg.rhomax) / cfg.cushion)
    Gr = s2 / rho + ms * (1 - th * th) / cfg.cushion
    GU = [-cfg.grp_beta[cfg.lig_group[l]] * cfg.lig_w[l] / (cfg.grp_alpha[cfg.lig_group[l]] + sum(cfg.lig_w[m] * ug[m + 1] for m in range(nlig) if cfg.lig_group[m] == cfg.lig_group[l])) for l in range(nlig)]
    L2 = symbol_D2(n, L / n)[None, :] + symbol_D2(n, L / n)[:, None]
    shift = 1 / (GAMMA * h)
    A = lambda v: shift * v - o.jvp(u, v)
    a_rr = np.mean(rho * Gr); a_rU = [np.mean(rho * g) for g in GU]
    def pc_plain(v):
        vh = np.fft.fft2(v.reshape(F, n, n))
        d = [shift + cfg.lig_gamma[l] - cfg.lig_D[l] * L2 for l in range(nlig)]
        den = shift - a_rr * L2 - sum(a_rU[l] * L2 * cfg.lig_s[l] / d[l] for l in range(nlig))
        z0 = (vh[0] + sum(a_rU[l] * L2 / d[l] * vh[l + 1] for l in range(nlig))) / den
        zs = [z0] + [(vh[l + 1] + cfg.lig_s[l] * z0) / d[l] for l in range(nlig)]
        return np.real(np.fft.ifft2(np.array(zs))).reshape(-1)
    sig = shift / (rho * Gr)
    Mk = np.zeros((n, n, F, F))
    Mk[..., 0, 0] = np.mean(sig) - L2
    for l in range(nlig):
        Mk[..., 0, l + 1] = np.mean(-sig * GU[l])
        Mk[..., l + 1, 0] = np.mean(-cfg.lig_s[l] / Gr)
        for m in range(nlig):
            Mk[..., l + 1, m + 1] = np.mean(cfg.lig_s[l] * GU[m] / Gr)
        Mk[..., l + 1, l + 1] += shift + cfg.lig_gamma[l] - cfg.lig_D[l] * L2
    Minv = np.linalg.inv(Mk)
    def pc_scaled(v):
        vv = v.reshape(F, n, n).copy()
        vv[0] = vv[0] / rho
        w = np.real(np.fft.ifft2(np.einsum('yxab,byx->ayx', Minv, np.fft.fft2(vv))))
        out = w.copy()
        out[0] = (w[0] - sum(GU[l] * w[l + 1] for l in range(nlig))) / Gr
        return out.reshape(-1)
    b = o.rhs(u); bn = np.linalg.norm(b)
    print('n=%d h=%g nlig=%d rho %.0f..%.0f' % (n, h, nlig, rho.min(), rho.max()))
    for name, pc in (('plain', pc_plain), ('scaled', pc_scaled)):
        x = np.zeros_like(b); r = b.copy(); hist = []
        for k in range(nsweep):
            x = x + pc(r); r = b - A(x); hist.append(np.linalg.norm(r) / bn)
        print('  %-6s ' % name + ' '.join('%.2e' % v for v in hist) + '   ratios ' + ' '.join('%.1e' % (hist[i + 1] / hist[i]) for i in range(len(hist) - 1)))


if __name__ == '__main__':
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    for h in (0.01, 0.2, 1.0, 10.0):
        run(n, h)
    run(n, 0.2, nlig=2)
    run(n, 0.2, amp=0.1)
