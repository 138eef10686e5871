"""CPU experiment: a spectral preconditioner for AGGREGATED states.  In the variables (dG, v_U), dG = G_rho v_rho + sum G_Ul v_Ul,
the stage matrix has constant-coefficient highest-order terms:
   rho row / rho :  sigma(x) (dG - sum G_U v_U) - Lap dG            sigma = shift / (rho G_rho)
   U rows        :  (shift + gamma - D Lap) v_U - (s/G_rho)(dG - sum G_U v_U)
so M = the same with the zeroth-order coefficient FIELDS replaced by their means is FFT-invertible, and the remainder is a bounded
(zeroth-order) perturbation plus the dropped gradient terms.  GMRES iteration counts on the late (aggregated) state."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
from ksfd_amd.config import ProblemConfig
from oracle import ko
import scipy.sparse.linalg as spl
from fft_pc_experiment import symbol_D2
GAMMA = 4.3586652150845900e-01

z = np.load('tools/_scratch/late_state.npz')
u = z['u']; n = 384; nlig = 2; F = 3
cfg = ProblemConfig.standard(2, (n, n), L=(1.0, 1.0), nlig=nlig)
o = ko.Oracle(cfg); N = cfg.N
ug = o.groom(u).reshape(F, n, n); rho = ug[0]
s2 = cfg.s2; ms = cfg.maxscale * s2
th = np.tanh((rho - cfg.rhomax) / cfg.cushion)
Gr = s2 / rho + ms * (1 - th * th) / cfg.cushion
GU = [-cfg.grp_beta[cfg.lig_group[l]] * cfg.lig_w[l] / (cfg.grp_alpha[cfg.lig_group[l]] + ug[l + 1]) for l in range(nlig)]
L2 = symbol_D2(n, 1.0 / n)[None, :] + symbol_D2(n, 1.0 / n)[:, None]
b = o.rhs(u)
print('rho %.0f..%.0f  rho*Grho %.2e..%.2e  rho*GU0 %.2e..%.2e' % (rho.min(), rho.max(), (rho * Gr).min(), (rho * Gr).max(), (rho * GU[0]).min(), (rho * GU[0]).max()))
for h in (1.0, 10.0, 100.0, float(z['h'])):
    shift = 1 / (GAMMA * h)
    A = lambda v: shift * v - o.jvp(u, v)
    # --- plain constant-coefficient M (means of rho*G_rho, rho*G_U)
    a_rr = np.mean(rho * Gr); a_rU = [np.mean(rho * g) for g in GU]
    def pc_plain(v):
        vh = np.fft.fft2(v.reshape(F, n, n))
        d = [shift + cfg.lig_gamma[l] - cfg.lig_D[l] * L2 for l in range(nlig)]
        den = shift - a_rr * L2 - sum(a_rU[l] * L2 * cfg.lig_s[l] / d[l] for l in range(nlig))
        z0 = (vh[0] + sum(a_rU[l] * L2 / d[l] * vh[l + 1] for l in range(nlig))) / den
        zs = [z0] + [(vh[l + 1] + cfg.lig_s[l] * z0) / d[l] for l in range(nlig)]
        return np.real(np.fft.ifft2(np.array(zs))).reshape(-1)
    # --- scaled: unknowns (dG, v_U); rows scaled by 1/rho (rho row)
    sig = shift / (rho * Gr)
    c00 = np.mean(sig)                         # sigma
    c0l = [np.mean(-sig * g) for g in GU]      # -sigma G_U
    cl0 = [np.mean(-cfg.lig_s[l] / Gr) for l in range(nlig)]
    cll = [[np.mean(cfg.lig_s[l] * GU[m] / Gr) for m in range(nlig)] for l in range(nlig)]
    Mk = np.zeros((n, n, F, F))
    Mk[..., 0, 0] = c00 - L2
    for l in range(nlig):
        Mk[..., 0, l + 1] = c0l[l]
        Mk[..., l + 1, 0] = cl0[l]
        for m in range(nlig):
            Mk[..., l + 1, m + 1] = cll[l][m]
        Mk[..., l + 1, l + 1] += shift + cfg.lig_gamma[l] - cfg.lig_D[l] * L2
    Minv = np.linalg.inv(Mk)
    def pc_scaled(v):
        vv = v.reshape(F, n, n).copy()
        vv[0] = vv[0] / rho
        wh = np.einsum('yxab,byx->ayx', Minv, np.fft.fft2(vv))
        w = np.real(np.fft.ifft2(wh))
        out = w.copy()
        out[0] = (w[0] - sum(GU[l] * w[l + 1] for l in range(nlig))) / Gr
        return out.reshape(-1)
    for name, pc in (('plain', pc_plain), ('scaled', pc_scaled)):
        cnt = [0]
        def cb(r): cnt[0] += 1
        x, info = spl.gmres(spl.LinearOperator((F * N, F * N), matvec=A), b, M=spl.LinearOperator((F * N, F * N), matvec=pc), rtol=1e-6, restart=40, maxiter=5, callback=cb, callback_type='pr_norm')
        print('  h=%-8g %-7s its %4d info %d  true rel res %.2e' % (h, name, cnt[0], info, np.linalg.norm(b - A(x)) / np.linalg.norm(b)))
