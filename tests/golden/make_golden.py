#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/*.npz by importing the REFERENCE's own
symbolic + generated-C operator layer (KSFD.Derivatives / StencilUfunc / SpatialExpression).

Runs ONLY in the build container (needs /root/reference, sympy, gcc); never on the GPU box.
The reference's PETSc/MPI/HDF5 dependencies are absent here, so four throw-away stand-in
packages in tests/golden/_stubs (single-rank mpi4py, permissive petsc4py, no-op dogpile
cache, empty h5py) are put ahead of the reference on sys.path.  They carry no arithmetic:
every number written below comes out of the reference's sympy expressions compiled by the
reference's own ufuncify (KSFD/ksfdufunc.py) with gcc.

What is captured per case:
  u, rhs=dfdt(u)            KSFD/ksfdsym.py:902-940 (drhodt :763-812, dUdt_ufs :615-628, groom :888-900)
  vel                       KSFD/ksfdsym.py:1188-1209
  Jrho_off/Jrho_val         rho-row Jacobian entry fields, KSFD/ksfdsym.py:675-761, 1067-1127
  JU_off/JU_val             U-row Jacobian constants,      KSFD/ksfdsym.py:630-673
  v, Jv                     J.v assembled from those entries with periodic wrap
  ug, rhs_g, vel_g          the same operators on an input with values < rhomin/Umin and NaNs
  step cases                reference RHS + reference Jacobian (scipy sparse, exact LU) driven by
                            this script's restatement of PETSc TS ROSW RA34PW2 (tableau in
                            SURVEY.md 8c; PETSc itself is not available -> "stepper parity unpinned")

Usage:  cd /tmp/somewhere && python /root/repo/tests/golden/make_golden.py [case ...]
(ufunc build products land in ./autowrap of the current directory.)
"""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(HERE, '_stubs'), '/root/reference']

import numpy as np                                   # noqa: E402
from argparse import Namespace                       # noqa: E402
import scipy.sparse as sp                            # noqa: E402
import scipy.sparse.linalg as spla                   # noqa: E402
import KSFD                                          # noqa: E402,F401
from KSFD import SolutionParameters, Derivatives     # noqa: E402
from KSFD.ksfdgrid import Grid                       # noqa: E402
from KSFD.ksfdsym import SpatialExpression           # noqa: E402

COMMON = ['sigma=0.02357', 's2=sigma**2/2', 'rhomax=28000', 'cushion=2000',
          'rhomin=1e-7', 'Umin=1e-7']
LIG_N1 = ['ngroups=1', 'nligands_1=1', 'alpha_1=1500', 'beta_1=5.56e-4', 's_1_1=0.01',
          'gamma_1_1=0.01', 'D_1_1=1e-6', 'U0_1_1=9000.0']
LIG_N2 = ['ngroups=2', 'nligands_1=1', 'alpha_1=1500', 'beta_1=5.56e-4', 's_1_1=0.01',
          'gamma_1_1=0.01', 'D_1_1=1e-6', 'U0_1_1=9000.0',
          'nligands_2=1', 'alpha_2=1500', 'beta_2=-5.56e-4', 's_2_1=0.001',
          'gamma_2_1=0.001', 'D_2_1=1e-5', 'U0_2_1=9000.0']
# one group with two ligands of different weights (exercises sum_l w_gl U_gl inside the log)
LIG_G2 = ['ngroups=1', 'nligands_1=2', 'alpha_1=1500', 'beta_1=5.56e-4',
          's_1_1=0.01', 'gamma_1_1=0.01', 'D_1_1=1e-6', 'weight_1_1=0.7', 'U0_1_1=9000.0',
          's_1_2=0.004', 'gamma_1_2=0.02', 'D_1_2=3e-6', 'weight_1_2=1.3', 'U0_1_2=9000.0']

MANUFACTURED_SRC = None   # filled from the reference's options93nx128dt1 on demand


class FakeGrid:
    """The few Grid attributes Derivatives/StencilUfunc/SpatialExpression read
    (KSFD/ksfdgrid.py:163-177, 413-434), single rank, no PETSc."""
    stencil_slice = Grid.stencil_slice

    def __init__(self, dim, n, L, dof):
        self.dim = dim
        self.stencil_width = 2
        self.nps = np.array(n[:dim], dtype=int)
        self.bounds = np.array(L[:dim], dtype=float)
        self.spacing = self.bounds / self.nps
        self.dof = dof
        self.Slshape = tuple(int(x) for x in n[:dim])
        self.Sashape = tuple(int(x) + 4 for x in n[:dim])
        self.Vlshape = (dof,) + self.Slshape
        self.Vashape = (dof,) + self.Sashape
        axes = [np.arange(n[d]) * (L[d] / n[d]) for d in range(dim)]
        mesh = np.meshgrid(*axes, indexing='ij')
        self.coordsNoGhosts = np.asfortranarray(np.stack(mesh))


def build(dim, n, L, lig, cap='tophat', extra=(), source=()):
    names = ['nwidth', 'nheight', 'ndepth']
    lens = ['width', 'height', 'depth']
    params = ['dim=%d' % dim]
    params += ['%s=%d' % (names[d], n[d]) for d in range(dim)]
    params += ['%s=%r' % (lens[d], float(L[d])) for d in range(dim)]
    params += COMMON + list(lig) + list(extra)
    cl = Namespace(params=params, cappotential=cap, source=list(source), petsc=[])
    ps = SolutionParameters(cl)
    F = ps.nligands + 1
    g = FakeGrid(dim, n, L, F)
    srcs = [SpatialExpression(ps, g, '0.0') for _ in range(F)]
    fnames = ['rho'] + [l.name() for l in ps.groups.ligands()]
    for s in source:
        k, val = s.split('=', 1)
        srcs[fnames.index(k)] = SpatialExpression(ps, g, val)
    d = Derivatives(ps, g, sources=srcs, u0=object())
    return ps, g, d


def fval(ps, x, t):
    """number behind a parameter: constants are plain numbers, time-dependent ones stay sympy symbols whose value the
    reference takes from ps.values(t) on every ufunc call (KSFD/ksfdsym.py:1303-1312, 1430-1439)"""
    try:
        return float(x)
    except TypeError:
        import sympy
        return float(sympy.sympify(x).subs({sympy.Symbol(k): val for k, val in ps.values(t).items()}))


def tables_at(ps, t):
    """the numeric ligand / group tables the operators see at time t"""
    ligs = list(ps.Vgroups.ligands())
    groups = ps.Vgroups.groups
    v = ps.values(t)
    return dict(
        s2=np.float64(v['s2']), rhomax=np.float64(v['rhomax']), cushion=np.float64(v['cushion']), maxscale=np.float64(v['maxscale']),
        lig_w=np.array([fval(ps, l.weight, t) for l in ligs]), lig_s=np.array([fval(ps, l.s, t) for l in ligs]),
        lig_gamma=np.array([fval(ps, l.gamma, t) for l in ligs]), lig_D=np.array([fval(ps, l.D, t) for l in ligs]),
        grp_alpha=np.array([fval(ps, gr.alpha, t) for gr in groups]), grp_beta=np.array([fval(ps, gr.beta, t) for gr in groups]))


def meta_of(ps, g, cap, t=None):
    v = ps.values0 if t is None else ps.values(t)
    ligs = list(ps.Vgroups.ligands())
    groups = ps.Vgroups.groups
    if t is not None:
        tb = tables_at(ps, t)
        n3 = [1, 1, 1]
        L3 = [1.0, 1.0, 1.0]
        for dd in range(g.dim):
            n3[dd] = int(g.nps[dd])
            L3[dd] = float(g.bounds[dd])
        return dict(tb, dim=np.int64(g.dim), n=np.array(n3, dtype=np.int64), L=np.array(L3), nlig=np.int64(len(ligs)),
                    ngroups=np.int64(len(groups)), cap_kind=np.int64(1 if cap == 'witch' else 0),
                    rhomin=np.float64(v['rhomin']), Umin=np.float64(v['Umin']),
                    lig_group=np.array([l.groupnum - 1 for l in ligs], dtype=np.int32))
    n3 = [1, 1, 1]
    L3 = [1.0, 1.0, 1.0]
    for dd in range(g.dim):
        n3[dd] = int(g.nps[dd])
        L3[dd] = float(g.bounds[dd])
    return dict(
        dim=np.int64(g.dim), n=np.array(n3, dtype=np.int64), L=np.array(L3),
        nlig=np.int64(len(ligs)), ngroups=np.int64(len(groups)),
        cap_kind=np.int64(1 if cap == 'witch' else 0),
        s2=np.float64(v['s2']), rhomax=np.float64(v['rhomax']), cushion=np.float64(v['cushion']),
        maxscale=np.float64(v['maxscale']), rhomin=np.float64(v['rhomin']), Umin=np.float64(v['Umin']),
        lig_group=np.array([l.groupnum - 1 for l in ligs], dtype=np.int32),
        lig_w=np.array([float(l.weight) for l in ligs]),
        lig_s=np.array([float(l.s) for l in ligs]),
        lig_gamma=np.array([float(l.gamma) for l in ligs]),
        lig_D=np.array([float(l.D) for l in ligs]),
        grp_alpha=np.array([float(gr.alpha) for gr in groups]),
        grp_beta=np.array([float(gr.beta) for gr in groups]),
    )


def ghosted(d, u):
    pad = ((0, 0),) + ((2, 2),) * d.dim
    farr = np.asfortranarray(np.pad(u, pad, mode='wrap'))
    return d.groom(farr)           # KSFD/ksfdsym.py:922 -- clamp applied to the ghosted copy


def ref_rhs(d, u, t=0.0):
    """KSFD/ksfdsym.py:902-940 without the PETSc Vec plumbing."""
    F = u.shape[0]
    farr = ghosted(d, u)
    out = np.zeros(u.shape, order='F')
    out[0] = d.drhodt(farr, t=t)
    out[0] += d.sources[0](t)
    for l, uf in enumerate(d.dUdt_ufs):
        tmp = np.empty(u.shape[1:], order='F')
        uf(farr, t=t, out=(tmp,))
        out[l + 1] = tmp + d.sources[l + 1](t)
    assert F == len(d.dUdt_ufs) + 1
    return out


def ref_vel(d, u, t=0.0):
    farr = ghosted(d, u)
    out = np.zeros((d.dim,) + u.shape[1:], order='F')
    for a, suf in enumerate(d.vel_ufuncs):
        tmp = np.empty(u.shape[1:], order='F')
        suf(farr, t=t, out=tmp)
        out[a] = tmp
    return out


def ref_jac_entries(d, u, t=0.0):
    """rho rows: KSFD/ksfdsym.py:675-761; U rows: :630-673."""
    farr = ghosted(d, u)
    S = u.shape[1:]
    acc = {}
    for suf in d.Jrhoufs:
        outs = tuple(np.empty(S, order='F') for _ in suf.out_stencils)
        suf(farr, t=t, out=outs)
        for o, st in zip(outs, suf.out_stencils):
            key = tuple(int(x) for x in st)
            acc[key] = acc.get(key, 0.0) + o
    keys = sorted(acc.keys(), key=lambda k: (k[3], k[2], k[1], k[0]))
    Jrho_off = np.array(keys, dtype=np.int64)
    Jrho_val = np.stack([acc[k] for k in keys])
    ssyms = list(d.stencil_sym_nums.keys())
    JU_off, JU_val, JU_row = [], [], []
    for lm1, lig in enumerate(d.ps.Vgroups.ligands()):
        e = d.dUdt_exp(lig, dof=lm1 + 1)
        args = e.free_symbols.intersection(ssyms)
        nums = sorted(d.stencil_sym_nums[s] for s in args)
        for nn in nums:
            JU_row.append(lm1 + 1)
            JU_off.append([int(x) for x in d.stencils[nn]])
            JU_val.append(float(e.diff(ssyms[nn]).subs(d.ps.values(t))))
    return (Jrho_off, Jrho_val, np.array(JU_row, dtype=np.int64),
            np.array(JU_off, dtype=np.int64), np.array(JU_val))


def shift(a, off, dim):
    """a evaluated at (i+di, j+dj, k+dk) with periodic wrap."""
    for ax in range(dim):
        if off[ax]:
            a = np.roll(a, -int(off[ax]), axis=ax)
    return a


def jv_from_entries(dim, ent, v):
    Jrho_off, Jrho_val, JU_row, JU_off, JU_val = ent
    out = np.zeros_like(v)
    for off, val in zip(Jrho_off, Jrho_val):
        out[0] += val * shift(v[off[3]], off, dim)
    for r, off, val in zip(JU_row, JU_off, JU_val):
        out[r] += val * shift(v[off[3]], off, dim)
    return out


def sparse_from_entries(dim, S, F, ent):
    """Unknown ordering: c*N + p, p = i + nx*(j + ny*k)  (SoA, x fastest)."""
    Jrho_off, Jrho_val, JU_row, JU_off, JU_val = ent
    N = int(np.prod(S))
    idx = np.arange(N).reshape(S, order='F')
    rows, cols, vals = [], [], []
    for off, val in zip(Jrho_off, Jrho_val):
        rows.append(idx.ravel(order='F'))
        cols.append(off[3] * N + shift(idx, off, dim).ravel(order='F'))
        vals.append(np.asarray(val).ravel(order='F'))
    for r, off, val in zip(JU_row, JU_off, JU_val):
        rows.append(r * N + idx.ravel(order='F'))
        cols.append(off[3] * N + shift(idx, off, dim).ravel(order='F'))
        vals.append(np.full(N, val))
    return sp.csc_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))),
                         shape=(F * N, F * N))


# --- PETSc TS ROSW RA34PW2, restated (PETSc src/ts/impls/rosw/rosw.c from memory) ---
GAM = 4.3586652150845900e-01
RA_A = np.array([[0, 0, 0, 0],
                 [8.7173304301691801e-01, 0, 0, 0],
                 [8.4457060015369423e-01, -1.1299064236484185e-01, 0, 0],
                 [0, 0, 1., 0]])
RA_G = np.array([[GAM, 0, 0, 0],
                 [-8.7173304301691801e-01, GAM, 0, 0],
                 [-9.0338057013044082e-01, 5.4180672388095326e-02, GAM, 0],
                 [2.4212380706095346e-01, -1.2232505839045147e+00, 5.4526025533510214e-01, GAM]])
RA_b = np.array([2.4212380706095346e-01, -1.2232505839045147e+00, 1.5452602553351020e+00, GAM])
RA_b2 = np.array([3.7810903145819369e-01, -9.6042292212423178e-02, 0.5, 2.1793326075422950e-01])


def rosw_step(d, u, t, h, atol, rtol):
    """One RA34PW2 step in PETSc's transformed variables Y = Gamma.k with an exact sparse LU.
    Returns (unew, err_vector, wrms_norm)."""
    dim, S, F = d.dim, u.shape[1:], u.shape[0]
    Ginv = np.linalg.inv(RA_G)
    At = RA_A @ Ginv
    bt = RA_b @ Ginv
    b2t = RA_b2 @ Ginv
    asum = RA_A.sum(axis=1)
    ent = ref_jac_entries(d, u, t)
    J = sparse_from_entries(dim, S, F, ent)
    shift_ = 1.0 / (GAM * h)
    lu = spla.splu((shift_ * sp.identity(J.shape[0], format='csc') - J).tocsc())
    flat = lambda a: np.concatenate([a[c].ravel(order='F') for c in range(F)])
    unflat = lambda x: np.stack([x[c * (x.size // F):(c + 1) * (x.size // F)].reshape(S, order='F')
                                 for c in range(F)])
    Y = []
    for i in range(4):
        Z = u.copy()
        Zdot = np.zeros_like(u)
        for j in range(i):
            Z = Z + At[i, j] * Y[j]
            Zdot = Zdot + (Ginv[i, j] / h) * Y[j]
        rhs = ref_rhs(d, Z, t + asum[i] * h) - Zdot
        Y.append(unflat(lu.solve(flat(rhs))))
    unew = u.copy()
    err = np.zeros_like(u)
    for j in range(4):
        unew = unew + bt[j] * Y[j]
        err = err + (b2t[j] - bt[j]) * Y[j]
    y_emb = unew + err
    tol = atol + rtol * np.maximum(np.abs(unew), np.abs(y_emb))
    wrms = float(np.sqrt(np.mean((err / tol) ** 2)))
    return unew, err, wrms


def seeded_state(rng, F, S, amp=900.0):
    return np.asfortranarray(9000.0 + amp * rng.standard_normal((F,) + tuple(S)))


def operator_case(name, dim, n, L, lig, cap, seed, witch_near_cap=False):
    t0 = time.time()
    ps, g, d = build(dim, n, L, lig, cap)
    F = ps.nligands + 1
    S = g.Slshape
    rng = np.random.default_rng(seed)
    u = seeded_state(rng, F, S)
    if witch_near_cap:
        # put part of rho near rhomax so tanh((rho-rhomax)/cushion) is not saturated
        u[0] = np.asfortranarray(24000.0 + 4000.0 * rng.standard_normal(S))
    out = meta_of(ps, g, cap)
    out['u'] = u
    out['rhs'] = ref_rhs(d, u)
    out['vel'] = ref_vel(d, u)
    ent = ref_jac_entries(d, u)
    out['Jrho_off'], out['Jrho_val'], out['JU_row'], out['JU_off'], out['JU_val'] = ent
    v = np.asfortranarray(rng.standard_normal((F,) + tuple(S)))
    out['v'] = v
    out['Jv'] = jv_from_entries(dim, ent, v)
    # groom-active input: negatives, values below the floors, NaNs  (KSFD/ksfdsym.py:888-900)
    ug = u.copy()
    flatidx = rng.choice(ug[0].size, size=max(4, ug[0].size // 16), replace=False)
    kinds = rng.integers(0, 3, size=flatidx.size)
    for c in range(F):
        plane = ug[c].reshape(-1, order='F').copy()
        sel = np.roll(flatidx, c)
        plane[sel[kinds == 0]] = -5.0
        plane[sel[kinds == 1]] = 1e-9
        plane[sel[kinds == 2]] = np.nan
        ug[c] = plane.reshape(S, order='F')
    out['ug'] = ug
    out['rhs_g'] = ref_rhs(d, ug)
    out['vel_g'] = ref_vel(d, ug)
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
    print('%-24s F=%d shape=%s  %.1fs' % (name, F, S, time.time() - t0), flush=True)


def step_case(name, dim, n, L, lig, cap, seed, h, nsteps, atol=0.01, rtol=1e-6, amp=90.0,
              source=(), extra=(), u0_fn=None, t0=0.0, time_dependent=False):
    tt = time.time()
    ps, g, d = build(dim, n, L, lig, cap, extra=extra, source=source)
    F = ps.nligands + 1
    S = g.Slshape
    rng = np.random.default_rng(seed)
    if u0_fn is None:
        u = seeded_state(rng, F, S, amp=amp)
        ligs = list(ps.Vgroups.ligands())
        for l, lg in enumerate(ligs):       # U = rho*s/gamma + small perturbation (ksfdsolver2.py:636-637)
            u[l + 1] = u[0] * fval(ps, lg.s / lg.gamma, t0) + 0.1 * amp * rng.standard_normal(S)
    else:
        u = u0_fn(g)
    out = meta_of(ps, g, cap, t=t0 if time_dependent else None)
    out['u0'] = u.copy()
    out['h'] = np.float64(h)
    out['t0'] = np.float64(t0)
    stage_tab = []
    out['atol'] = np.float64(atol)
    out['rtol'] = np.float64(rtol)
    out['nsteps'] = np.int64(nsteps)
    t = t0
    wr = []
    has_src = len(source) > 0
    if has_src:
        asum = RA_A.sum(axis=1)
        src_t, src_v = [], []
    for k in range(nsteps):
        if has_src:
            for i in range(4):
                ts_ = t + asum[i] * h
                src_t.append(ts_)
                src_v.append(np.stack([np.broadcast_to(d.sources[c](ts_), S) + 0.0 for c in range(F)]))
        if time_dependent:          # what ps.values() hands the ufuncs: at t_n for the Jacobian, at the stage times for the RHS
            stage_tab.append([tables_at(ps, t)] + [tables_at(ps, t + a * h) for a in RA_A.sum(axis=1)])
        unew, err, wrms = rosw_step(d, u, t, h, atol, rtol)
        if k == 0:
            out['u1'] = unew.copy()
            out['err1'] = err.copy()
        wr.append(wrms)
        u = unew
        t += h
    out['uN'] = u
    out['wrms'] = np.array(wr)
    if time_dependent:
        for key in stage_tab[0][0]:        # (nsteps, 5, ...): index 0 = step start (Jacobian), 1..4 = the four stage times
            out['tdep_' + key] = np.array([[tb[key] for tb in row] for row in stage_tab])
    if has_src:
        out['src_t'] = np.array(src_t)
        out['src_v'] = np.stack(src_v)          # (4*nsteps, F, *S): source fields at every stage time
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
    print('%-24s F=%d shape=%s h=%g steps=%d wrms[0]=%.3e  %.1fs' %
          (name, F, S, h, nsteps, wr[0], time.time() - tt), flush=True)


def manufactured_source():
    """The --source=rho=... expression and constants of the reference's known-answer input
    (options93nx128dt1:22, 39-46), read from the reference at generation time."""
    import shlex
    src, extra = None, []
    for line in open('/root/reference/options93nx128dt1'):
        toks = shlex.split(line, comments=True)
        for tk in toks:
            if tk.startswith('--source='):
                src = tk[len('--source='):]
            elif tk.split('=')[0] in ('murho', 'arho', 'aUa', 'aUr', 'lamda', 'k0'):
                extra.append(tk)
    return src, extra


def exact93(g, t, consts):
    x = g.coordsNoGhosts[0]
    sn = np.sin(2 * np.pi * (0.25 + consts['k0'] * x))
    e = consts['arho'] * np.exp(consts['lamda'] * t)
    return np.asfortranarray(np.stack([consts['murho'] + e * sn,
                                       consts['murho'] + e * consts['aUa'] * sn,
                                       consts['murho'] + e * consts['aUr'] * sn]))


CASES = {}


def case(fn):
    CASES[fn.__name__] = fn
    return fn


@case
def op_1d_n1():
    operator_case('op_1d_n1', 1, [32], [1.0], LIG_N1, 'tophat', 11)


@case
def op_2d_n1():
    operator_case('op_2d_n1', 2, [16, 16], [1.0, 1.0], LIG_N1, 'tophat', 12)


@case
def op_2d_n2_aniso():
    operator_case('op_2d_n2_aniso', 2, [24, 40], [4.0, 2.5], LIG_N2, 'tophat', 13)


@case
def op_2d_n1_witch():
    operator_case('op_2d_n1_witch', 2, [16, 20], [1.0, 1.5], LIG_N1, 'witch', 14, witch_near_cap=True)


@case
def op_2d_n1_cap():
    operator_case('op_2d_n1_cap', 2, [20, 16], [0.5, 1.0], LIG_N1, 'tophat', 15, witch_near_cap=True)


@case
def op_2d_g2():
    operator_case('op_2d_g2', 2, [16, 12], [1.0, 1.0], LIG_G2, 'tophat', 16)


@case
def op_3d_n1():
    operator_case('op_3d_n1', 3, [12, 10, 8], [1.0, 1.2, 0.8], LIG_N1, 'tophat', 17)


@case
def op_3d_n2_witch():
    operator_case('op_3d_n2_witch', 3, [8, 8, 10], [1.0, 1.0, 1.0], LIG_N2, 'witch', 18)


@case
def step_2d_n1_mild():
    step_case('step_2d_n1_mild', 2, [16, 16], [1.0, 1.0], LIG_N1, 'tophat', 21, h=1.0, nsteps=10)


@case
def step_2d_n1_stiff():
    # small box -> hγ·λmax(s2·Δ) ≈ 0.436·0.05·(2·16/3·(16/0.02)²·2.78e-4) ≈ 41
    step_case('step_2d_n1_stiff', 2, [16, 16], [0.02, 0.02], LIG_N1, 'tophat', 22, h=0.05, nsteps=10)


@case
def step_2d_n2():
    step_case('step_2d_n2', 2, [16, 12], [0.1, 0.1], LIG_N2, 'tophat', 23, h=0.2, nsteps=5)


@case
def step_1d_n1():
    step_case('step_1d_n1', 1, [64], [0.05], LIG_N1, 'tophat', 24, h=0.1, nsteps=10)


@case
def step_3d_n1():
    step_case('step_3d_n1', 3, [8, 8, 8], [0.05, 0.05, 0.05], LIG_N1, 'tophat', 25, h=0.1, nsteps=3)


def adaptive_case(name, dim, n, L, lig, cap, seed, nsteps, dt0=1e-8, atol=0.01, rtol=1e-6, amp=90.0, near_cap=False):
    """TSAdaptBasic restated: accept iff wrms<=1; h_next = h*clip(safety*wrms^(-1/3), 0.1, 5), safety
    0.9 (x0.5 after a rejection); reference operators + exact LU per attempt.  Records every accepted (t,h) and wrms."""
    tt = time.time()
    ps, g, d = build(dim, n, L, lig, cap)
    F = ps.nligands + 1
    S = g.Slshape
    rng = np.random.default_rng(seed)
    u = seeded_state(rng, F, S, amp=amp)
    if near_cap:
        u[0] = np.asfortranarray(23000.0 + 2500.0 * np.sin(2 * np.pi * g.coordsNoGhosts[0] / L[0]) *
                                 np.cos(2 * np.pi * g.coordsNoGhosts[1] / L[1]) + amp * rng.standard_normal(S))
    for l, lg in enumerate(ps.Vgroups.ligands()):
        u[l + 1] = u[0] * float(lg.s / lg.gamma)
    out = meta_of(ps, g, cap)
    out['u0'] = u.copy()
    out['dt0'], out['atol'], out['rtol'] = np.float64(dt0), np.float64(atol), np.float64(rtol)
    t, h = 0.0, dt0
    ts, hs, wr, rejs = [], [], [], []
    for k in range(nsteps):
        prev_accept, nrej = True, 0
        while True:
            unew, err, w = rosw_step(d, u, t, h, atol, rtol)
            safety = 0.9
            accept = w <= 1.0
            if not accept and not prev_accept:
                safety *= 0.5
            hfac = min(max(safety * w ** (-1.0 / 3.0), 0.1), 5.0) if w > 0 else 5.0
            hnext = min(max(h * hfac, 1e-20), 1e4)
            prev_accept = accept
            if accept:
                u, t = unew, t + h
                ts.append(t); hs.append(h); wr.append(w); rejs.append(nrej)
                h = hnext
                break
            nrej += 1
            h = hnext
    out['uN'] = u
    out['t_acc'], out['h_acc'], out['wrms'], out['rej'] = np.array(ts), np.array(hs), np.array(wr), np.array(rejs)
    out['h_next'] = np.float64(h)
    out['nsteps'] = np.int64(nsteps)
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
    print('%-24s F=%d shape=%s steps=%d t_end=%.4g h_end=%.3g rejections=%d  %.1fs' %
          (name, F, S, nsteps, t, h, int(np.sum(rejs)), time.time() - tt), flush=True)


@case
def adapt_2d_n1():
    adaptive_case('adapt_2d_n1', 2, [16, 16], [0.04, 0.04], LIG_N1, 'tophat', 31, nsteps=24)


@case
def adapt_2d_n1_reject():
    # first trial step far too large: several consecutive rejections (reject_safety path), then recovery
    adaptive_case('adapt_2d_n1_reject', 2, [16, 16], [0.04, 0.04], LIG_N1, 'tophat', 33, nsteps=10, dt0=2.0)


@case
def adapt_2d_n2_witch_cap():
    adaptive_case('adapt_2d_n2_witch_cap', 2, [16, 12], [0.05, 0.04], LIG_N2, 'witch', 32, nsteps=16, near_cap=True)


@case
def step_2d_n1_witch_cap():
    # rho up to ~rhomax: the cap potential's tanh is active in G, G_rho
    def u0(g):
        rng = np.random.default_rng(41)
        x, y = g.coordsNoGhosts[0], g.coordsNoGhosts[1]
        rho = 24000.0 + 3500.0 * np.sin(2 * np.pi * x / 0.05) * np.cos(2 * np.pi * y / 0.05) + 9 * rng.standard_normal(g.Slshape)
        return np.asfortranarray(np.stack([rho, rho]))
    step_case('step_2d_n1_witch_cap', 2, [16, 16], [0.05, 0.05], LIG_N1, 'witch', 41, h=0.002, nsteps=5, u0_fn=u0)


@case
def step_2d_g2():
    step_case('step_2d_g2', 2, [12, 16], [0.04, 0.05], LIG_G2, 'tophat', 42, h=0.1, nsteps=4)


@case
def step_2d_n1_tdep():
    """time-dependent coefficients: s_1_1 and beta_1 are expressions of t, so every RHS evaluation sees ps.values(t_stage)
    and the Jacobian ps.values(t_n) (KSFD/ksfdsym.py:1303-1312, 1430-1439; KSFD/ksfdts.py:563-640)"""
    lig = ['ngroups=1', 'nligands_1=1', 'alpha_1=1500', 'beta_1=5.56e-4*(1+0.3*t)', 's_1_1=0.01*(1+0.8*t)',
           'gamma_1_1=0.01', 'D_1_1=1e-6', 'U0_1_1=9000.0']
    step_case('step_2d_n1_tdep', 2, [20, 16], [0.06, 0.05], lig, 'tophat', 31, h=0.1, nsteps=3, t0=0.5, time_dependent=True)


@case
def step_1d_manufactured():
    src, extra = manufactured_source()
    consts = {kv.split('=')[0]: float(kv.split('=')[1]) for kv in extra}
    lig = [x for x in LIG_N2 if not x.startswith('U0_')] + ['U0_1_1=9000.0', 'U0_2_1=9000.0']
    step_case('step_1d_manufactured', 1, [128], [1.0], lig, 'tophat', 26, h=1.0, nsteps=20,
              source=[src], extra=extra, u0_fn=lambda g: exact93(g, 0.0, consts))
    f = os.path.join(HERE, 'step_1d_manufactured.npz')
    z = dict(np.load(f))
    ps, g, d = build(1, [128], [1.0], lig, 'tophat', extra=extra)
    z['exactN'] = exact93(g, 20.0, consts)
    for k, v in consts.items():
        z['const_' + k] = np.float64(v)
    np.savez_compressed(f, **z)
    print('   manufactured: max|uN-exact| =', float(np.abs(z['uN'] - z['exactN']).max()))


if __name__ == '__main__':
    want = sys.argv[1:] or list(CASES)
    for w in want:
        CASES[w]()
