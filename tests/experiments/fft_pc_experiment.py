"""CPU experiment (not a test): GMRES iterations on the oracle's Jacobian with a constant-coefficient FFT
preconditioner (mean-coefficient symbol of shift*I - J inverted per wavenumber) vs none, bench initial data."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
from ksfd_amd.config import ProblemConfig
from ksfd_amd.initial import start_values
from oracle import ko
import scipy.sparse.linalg as spl

GAMMA = 4.3586652150845900e-01

def symbol_D2(n, h):
    th = 2 * np.pi * np.fft.fftfreq(n)
    return (-30 + 32 * np.cos(th) - 2 * np.cos(2 * th)) / (12 * h * h)

def run(n, hs, state=None, nlig=1, amp=None):
    L = n * 4.0 / 1536
    cfg = ProblemConfig.standard(2, (n, n), L=(L, L), nlig=nlig)
    o = ko.Oracle(cfg)
    u = start_values(cfg) if state is None else state
    if amp is not None:
        rng = np.random.default_rng(1)
        u = u.reshape(cfg.F, -1).copy()
        u[0] *= np.exp(amp * rng.standard_normal(u[0].size))
        u = u.reshape(-1)
    F, N = cfg.F, cfg.N
    ug = o.groom(u).reshape(F, n, n)
    rho = ug[0]
    # coefficient means
    eps = 1e-3
    # G_rho, G_U by closed form (tophat)
    s2 = cfg.s2; ms = cfg.maxscale * s2
    th = np.tanh((rho - cfg.rhomax) / cfg.cushion)
    Grho = s2 / rho + ms * (1 - th * th) / cfg.cushion
    GU = []
    for l in range(nlig):
        g = cfg.lig_group[l]
        ssum = cfg.grp_alpha[g] + sum(cfg.lig_w[m] * ug[m + 1] for m in range(nlig) if cfg.lig_group[m] == g)
        GU.append(-cfg.grp_beta[g] * cfg.lig_w[l] / ssum)
    a_rr = float(np.mean(rho * Grho)); a_rU = [float(np.mean(rho * g)) for g in GU]
    hx = L / n
    L2 = symbol_D2(n, hx)[None, :] + symbol_D2(n, hx)[:, None]
    print('n=%d rho range %.0f..%.0f a_rr %.3e a_rU %s' % (n, rho.min(), rho.max(), a_rr, a_rU))
    rng = np.random.default_rng(0)
    b = o.rhs(u)
    for h in hs:
        shift = 1 / (GAMMA * h)
        M = np.zeros((n, n, F, F), dtype=np.float64)
        M[..., 0, 0] = shift - a_rr * L2
        for l in range(nlig):
            M[..., 0, l + 1] = -a_rU[l] * L2
            M[..., l + 1, 0] = -cfg.lig_s[l]
            M[..., l + 1, l + 1] = shift + cfg.lig_gamma[l] - cfg.lig_D[l] * L2
        Minv = np.linalg.inv(M)
        def pc(v):
            vh = np.fft.fft2(v.reshape(F, n, n))
            out = np.einsum('yxab,byx->ayx', Minv, vh)
            return np.real(np.fft.ifft2(out)).reshape(-1)
        def A(v):
            return shift * v - o.jvp(u, v)
        cnt = {'n': 0}
        def cb(r): cnt['n'] += 1
        res = {}
        for name, Mop in (('fft', spl.LinearOperator((F * N, F * N), matvec=pc)), ('none', None)):
            if name == 'none' and h > 0.3: continue
            cnt['n'] = 0
            x, info = spl.gmres(spl.LinearOperator((F * N, F * N), matvec=A), b, M=Mop, rtol=1e-6, restart=30, maxiter=20, callback=cb, callback_type='pr_norm')
            res[name] = (cnt['n'], info, np.linalg.norm(b - A(x)) / np.linalg.norm(b))
        print('  h=%g shift=%.3g: %s' % (h, shift, res))

if __name__ == '__main__':
    run(256, [0.01, 0.2, 1.0, 10.0, 100.0])
    run(256, [0.2, 10.0], amp=0.1)
    run(256, [0.2, 10.0], amp=0.5)
    run(256, [0.2, 10.0], nlig=2)
    if os.path.exists('tools/_scratch/late_state.npz'):
        z = np.load('tools/_scratch/late_state.npz')
        print({k: z[k].shape for k in z})

def run_late():
    z = np.load('tools/_scratch/late_state.npz')
    u = z['u']; n = 384
    F = u.size // (n * n)
    print('late state t=%g h=%g F=%d' % (float(z['t']), float(z['h']), F))
    return u, n, F
