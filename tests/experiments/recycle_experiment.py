"""CPU experiment behind DESIGN.md 4c (not a test): FGMRES + Chebyshev polynomial on the oracle's assembled Jacobian, with and
without projecting each stage's right-hand side on the Arnoldi spaces of earlier stages.
  python tests/experiments/recycle_experiment.py [n] [h]   env: RMODE=all|last|first|first2|sel  KTRUNC=k  RTOL1=tol"""
import os, sys, numpy as np, scipy.sparse as sp
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from bench import build_problem
from ksfd_amd.initial import start_values
from oracle import ko
n = int(sys.argv[1]) if len(sys.argv) > 1 else 192
h = float(sys.argv[2]) if len(sys.argv) > 2 else 0.009
cfg = build_problem(n, 1)
u = start_values(cfg)
o = ko.Oracle(cfg)
F, N = cfg.F, cfg.N
At, Gi, bt, b2t, asum = ko.tableau()
GAM = 0.43586652150845900
def cheb_coeffs(a, b, target=0.02, maxdeg=6):
    kappa = b / a
    rc = (np.sqrt(kappa) - 1) / (np.sqrt(kappa) + 1)
    d = int(np.ceil(np.log(target) / np.log(rc))) - 1
    d = min(max(d, 1), maxdeg)
    nn = d + 1
    m0, m1 = (b + a) / (b - a), -2.0 / (b - a)
    Tp = np.poly1d([1.0]); Tc = np.poly1d([m1, m0])
    for k in range(1, nn):
        Tp, Tc = Tc, 2 * np.poly1d([m1, m0]) * Tc - Tp
    c = Tc.coeffs[::-1] / Tc(0.0)        # r(l) coefficients ascending
    return -c[1:], d                      # p_i = -r_{i+1}
def fgmres(A, shift, b, p, x0=None, rtol=1e-6, maxit=30):
    # returns x, its, V, Z, Hraw
    nrm_b = np.linalg.norm(b)
    x = np.zeros_like(b) if x0 is None else x0.copy()
    r = b - A @ x if x0 is not None else b.copy()
    beta = np.linalg.norm(r)
    V = [r / beta]; Z = []; H = np.zeros((maxit + 1, maxit))
    res = beta
    for j in range(maxit):
        v = V[j]
        # z = p(A/shift) v / shift  (Horner)
        z = p[-1] * v
        for c in p[-2::-1]:
            z = c * v + (A @ z) / shift
        z = z / shift
        w = A @ z
        for i in range(j + 1):
            H[i, j] = V[i] @ w; w = w - H[i, j] * V[i]
        for i in range(j + 1):
            c = V[i] @ w; H[i, j] += c; w = w - c * V[i]
        H[j + 1, j] = np.linalg.norm(w)
        V.append(w / H[j + 1, j]); Z.append(z)
        e1 = np.zeros(j + 2); e1[0] = beta
        y, *_ = np.linalg.lstsq(H[:j + 2, :j + 1], e1, rcond=None)
        res = np.linalg.norm(e1 - H[:j + 2, :j + 1] @ y)
        if res <= rtol * nrm_b:
            break
    x = x + np.column_stack(Z) @ y
    return x, j + 1, V, Z, H[:j + 2, :j + 1].copy(), res / nrm_b
def topetsc(x): return x.reshape(F, N).T.reshape(-1)
def tosoa(x): return x.reshape(N, F).T.reshape(-1)
for step in range(3):
    u = o.groom(u)
    rowptr, col, val = o.jacobian_csr(u)
    J = sp.csr_matrix((val, col, rowptr), shape=(F * N, F * N))
    shift = 1.0 / (GAM * h)
    A = (shift * sp.identity(F * N, format='csr') - J).tocsr()
    # lambda max by power iteration
    v = np.random.default_rng(0).standard_normal(F * N); v /= np.linalg.norm(v)
    for _ in range(12):
        w = A @ v; lam = np.linalg.norm(w); v = w / lam
    lamJ = lam - shift
    p, d = cheb_coeffs(0.97, 1 + 1.15 * lamJ / shift)
    Y = []; spaces = []
    its0 = []; its1 = []; red = []
    for i in range(4):
        zin = u + sum(At[i][j] * Y[j] for j in range(i))
        b = o.rhs(zin) - sum(Gi[i][j] / h * Y[j] for j in range(i))
        bp = topetsc(b)
        x_a, k_a, V, Z, H, rel = fgmres(A, shift, bp, p, rtol=(float(os.environ.get('RTOL1','1e-6')) if i == 0 else 1e-6))
        its0.append(k_a)
        # recycled: sequential projection on earlier spaces
        x0 = np.zeros_like(bp); r = bp.copy()
        import os
        mode=os.environ.get('RMODE','all')
        use = (spaces[:1] if i < 3 else [spaces[0], spaces[2]]) if mode=='sel' else spaces if mode=='all' else (spaces[-1:] if mode=='last' else (spaces[:1] if mode=='first' else spaces[:2]))
        for (Vs, Zs, Hs) in use:
            kk=int(os.environ.get('KTRUNC','99')); kk=min(kk,Hs.shape[1]); Hs=Hs[:kk+1,:kk]; Zs=Zs[:kk]
            Vm = np.column_stack(Vs[:Hs.shape[0]]); g = Vm.T @ r
            y, *_ = np.linalg.lstsq(Hs, g, rcond=None)
            x0 += np.column_stack(Zs) @ y
            r -= Vm @ (Hs @ y)
        red.append(np.linalg.norm(r) / np.linalg.norm(bp))
        if spaces:
            x_b, k_b, V, Z, H, rel = fgmres(A, shift, bp, p, x0=x0)
        else:
            x_b, k_b = x_a, k_a
        its1.append(k_b)
        spaces.append((V, Z, H))
        Y.append(tosoa(x_b))
    u = u + sum(bt[j] * Y[j] for j in range(4))
    print('step', step, 'X', (16/3)*2*(n/ (4.0*n/1536))**2 * 1e-6 / shift if False else '', 'deg', d, 'its x0=0', its0, 'recycled', its1, 'proj residual', ['%.1e' % x for x in red], flush=True)
