"""Command-line / @options-file front end with the reference's syntax (ksfdsolver2.py:33-52, 380-422;
KSFD/ksfdargparse.py:60-128), so the same `@options` files drive the HIP stepper:

  * `@file` indirection, `#` comments, shell-style quoting (shlex)        ksfdargparse.py:92-94
  * `--petsc ... --` pass-through block                                    ksfdargparse.py:106-128
  * `--option=value` switches                                              ksfdsolver2.py:380-422
  * free-form `name=sympy-expression` parameters that may depend on each other and on t, resolved
    through their dependency graph                                          KSFD/ksfdsoln.py:254-347

Only what the hot path consumes is interpreted: grid/box, tolerances, ligand tables, cap potential,
sources, the TSAdapt options inside the --petsc block.  The reference's own Parser/SolutionParameters
stay the authority; this module exists because the reference cannot be imported on the GPU box.
"""
import argparse
import math
import shlex

import numpy as np
import sympy as sy

from .config import ProblemConfig
from .lib import default_step_opts

# KSFD/ksfdargparse.py:11-55 (name, default) -- ligand/group defaults that actually take effect are 1.0
# (KSFD/ksfdligand.py:578-600; SURVEY.md section 5 "Gotcha")
DEFAULTS = dict(
    degree=3, dim=1, nelements=8, randgridnw=0, randgridnh=0, randgridnd=0,
    width=1.0, height=1.0, depth=1.0, CFL_safety_factor=0.0, conserve_worms=False,
    variance_rate=0.0, variance_interval=100.0, variance_timing_function='t/variance_interval',
    Umin=1e-7, rhomin=1e-7, rhomax=28000, cushion=2000, maxscale=2.0, s2=5.56e-4, Nworms=0.0,
    srho0=90.0, rho0=9000.0, ngroups=1, maxsteps=1000, t0=0.0, dt=0.001, lastvart=0.0, tmax=200000,
    rtol=1e-5, atol=1e-5,
)
GROUP_DEFAULTS = dict(alpha=1.0, beta=1.0, nligands=1)
LIGAND_DEFAULTS = dict(weight=1.0, s=1.0, gamma=1.0, D=1.0, series=1, depth=0.4)


def read_args(argv):
    """Expand @files (recursively) the way argparse's fromfile_prefix_chars + shlex(comments) does."""
    out = []
    for a in argv:
        if a.startswith('@'):
            with open(a[1:]) as f:
                for line in f:
                    out.extend(read_args(shlex.split(line, comments=True)))
        else:
            out.append(a)
    return out


def split_petsc(args):
    args = list(args)
    petsc = []
    while '--petsc' in args:
        f = args.index('--petsc')
        try:
            e = args.index('--', f + 1)
        except ValueError:
            e = len(args)
        petsc += args[f + 1:e]
        args[f:e + 1] = []
    return args, petsc


def parse_commandline(argv):
    args, petsc = split_petsc(read_args(argv))
    p = argparse.ArgumentParser(description='Solve Keller-Segel PDEs (MI355X HIP stepper)', allow_abbrev=False)
    p.add_argument('--cappotential', choices=['tophat', 'witch'], default='tophat')
    p.add_argument('--save')
    p.add_argument('--check')
    p.add_argument('--async_save', action='store_true')      # extension: write the series from a background thread
    p.add_argument('--saveevery', type=int, default=1)       # extension: keep every N-th step in the series
    p.add_argument('--resume')
    p.add_argument('--restart')
    p.add_argument('--series_retries', type=int, default=0)
    p.add_argument('--series_retry_interval', type=int, default=60)
    p.add_argument('--mpiok', action='store_true')
    p.add_argument('--showparams', action='store_true')
    p.add_argument('--noperiodic', action='store_true')
    p.add_argument('--onestep', action='store_true')
    p.add_argument('--solver', default='hip')
    p.add_argument('--seed', type=int, default=793817931)
    p.add_argument('--source', type=str, action='append', default=[])
    p.add_argument('params', type=str, nargs='*')
    ns = p.parse_intermixed_args(args)          # name=value parameters and --options may be interleaved
    ns.petsc = petsc
    return ns


def _sympify(val):
    if isinstance(val, str):
        if val == '':
            return None
        return sy.sympify(val, locals={'lamda': sy.Symbol('lamda'), 'beta': sy.Symbol('beta'),
                                       'gamma': sy.Symbol('gamma'), 'S': sy.Symbol('S')})
    return sy.sympify(val)


class Params:
    """Numeric view of the parameter set: Params.values(t) plays the role of ps.values(t)."""

    def __init__(self, clargs):
        self.clargs = clargs
        raw = dict(DEFAULTS)
        given = {}
        for arg in clargs.params:
            k, v = arg.split('=', 1)
            if k in given:
                raise ValueError('duplicated parameters: ' + k)
            given[k] = v
        raw.update(given)
        # group / ligand parameters that were not given
        ng = int(_sympify(raw.get('ngroups', 1)))
        for g in range(1, ng + 1):
            for k, d in GROUP_DEFAULTS.items():
                raw.setdefault('%s_%d' % (k, g), d)
            nl = int(_sympify(raw['nligands_%d' % g]))
            for l in range(1, nl + 1):
                for k, d in LIGAND_DEFAULTS.items():
                    raw.setdefault('%s_%d_%d' % (k, g, l), d)
                raw.setdefault('U0_%d_%d' % (g, l), '')
        for k in ('nwidth', 'nheight', 'ndepth'):
            raw.setdefault(k, raw['nelements'])
        self.raw = raw
        self.given = given
        self._resolve()

    def _resolve(self):
        t, x, y, z = sy.symbols('t x y z')
        leaves = {t, x, y, z}
        exprs = {}
        for k, v in self.raw.items():
            e = _sympify(v) if not isinstance(v, bool) else v
            exprs[k] = e
        done = {}
        pending = dict(exprs)
        while pending:
            progressed = False
            for k in list(pending):
                e = pending[k]
                if e is None or isinstance(e, bool) or not hasattr(e, 'free_symbols'):
                    done[k] = e
                    del pending[k]
                    progressed = True
                    continue
                deps = {str(s) for s in e.free_symbols - leaves}
                if deps - set(exprs):
                    raise ValueError('parameter %s depends on unknown symbol(s) %s' % (k, sorted(deps - set(exprs))))
                if deps & set(pending) - {k}:
                    continue
                if k in deps:
                    raise ValueError('parameter %s depends on itself' % k)
                done[k] = e.subs({sy.Symbol(d): done[d] for d in deps})
                del pending[k]
                progressed = True
            if not progressed:
                raise ValueError('cyclic parameter dependencies: ' + ', '.join(sorted(pending)))
        self.exprs = done
        self.t0 = float(done['t0'])
        self.values0 = self.values(self.t0)
        self.params0 = dict(self.values0)

    def values(self, t=None):
        t = self.t0 if t is None else t
        out = {}
        for k, e in self.exprs.items():
            if e is None or isinstance(e, bool):
                out[k] = e
                continue
            if not e.free_symbols:
                out[k] = _num(e)
            elif e.free_symbols == {sy.Symbol('t')}:
                out[k] = _num(e.subs({sy.Symbol('t'): t}))
            else:
                out[k] = e.subs({sy.Symbol('t'): t})        # spatial expression (rho0, U0_*)
        out['t'] = t
        return out

    def time_dependent(self):
        return sorted(k for k, e in self.exprs.items()
                      if hasattr(e, 'free_symbols') and sy.Symbol('t') in e.free_symbols)

    # ---- views used by the solver
    @property
    def dim(self):
        return int(self.values0['dim'])

    @property
    def shape(self):
        v = self.values0
        return tuple(int(v[k]) for k in ('nwidth', 'nheight', 'ndepth'))[:self.dim]

    @property
    def box(self):
        v = self.values0
        return tuple(float(v[k]) for k in ('width', 'height', 'depth'))[:self.dim]

    def ligands(self, t=None):
        """The already-expanded ligand table (after fourier_series, KSFD/ksfdligand.py:315-388):
        list of dicts(group, num, weight, s, gamma, D)."""
        v = self.values(t)
        out = []
        for g in range(1, int(v['ngroups']) + 1):
            num = 0
            for l in range(1, int(v['nligands_%d' % g]) + 1):
                base = {k: float(v['%s_%d_%d' % (k, g, l)]) for k in ('weight', 's', 'gamma', 'D', 'depth')}
                n = int(round(float(v['series_%d_%d' % (g, l)])))
                terms = []
                for i in range(n):
                    omega = math.pi * i / base['depth']
                    terms.append(dict(weight=base['weight'] / n, s=base['s'] / n,
                                      gamma=base['gamma'] + base['D'] * omega ** 2, D=base['D']))
                single = base['s'] / base['gamma']
                series = sum(tm['s'] / tm['gamma'] for tm in terms)
                for tm in terms:
                    tm['s'] *= single / series
                    num += 1
                    out.append(dict(group=g - 1, num=num, **tm))
        return out

    def problem_config(self, t=None):
        v = self.values(t)
        ligs = self.ligands(t)
        ng = int(v['ngroups'])
        return ProblemConfig(
            dim=self.dim, n=self.shape, L=self.box,
            lig_group=[l['group'] for l in ligs], lig_w=[l['weight'] for l in ligs],
            lig_s=[l['s'] for l in ligs], lig_gamma=[l['gamma'] for l in ligs], lig_D=[l['D'] for l in ligs],
            grp_alpha=[float(v['alpha_%d' % (g + 1)]) for g in range(ng)],
            grp_beta=[float(v['beta_%d' % (g + 1)]) for g in range(ng)],
            s2=float(v['s2']), rhomax=float(v['rhomax']), cushion=float(v['cushion']),
            maxscale=float(v['maxscale']), rhomin=float(v['rhomin']), Umin=float(v['Umin']),
            cap_kind=1 if self.clargs.cappotential == 'witch' else 0)

    def field_names(self):
        names, v = ['rho'], self.values0
        for g in range(1, int(v['ngroups']) + 1):
            num = 0
            for l in range(1, int(v['nligands_%d' % g]) + 1):
                for _ in range(int(round(float(v['series_%d_%d' % (g, l)])))):
                    num += 1
                    names.append('U_%d_%d' % (g, num))
        return names


def _num(e):
    f = float(e)
    return int(f) if getattr(e, 'is_Integer', False) else f


class SpatialExpression:
    """Counterpart of KSFD.SpatialExpression (KSFD/ksfdsym.py:1515-1697): a sympy expression of
    x, y, z, t and parameters, evaluated on the grid points x_i = i*L/n with numpy (host side)."""

    def __init__(self, params, expression='0.0'):
        self.params = params
        self.expression = _sympify(expression) if isinstance(expression, str) else sy.sympify(expression)
        syms = sy.symbols('x y z t')
        self._known = set(syms)
        self._fn = None

    def __call__(self, t, coords):
        """coords: list of dim arrays (broadcastable); returns array (or scalar broadcast by caller)."""
        vals = self.params.values(t)
        e = self.expression
        subs = {}
        for s in e.free_symbols - self._known:
            if str(s) not in vals:
                raise ValueError('unknown symbol(s) [%s]' % s)
            subs[s] = vals[str(s)]
        e = e.subs(subs)
        x, y, z, tt = sy.symbols('x y z t')
        f = sy.lambdify((x, y, z, tt), e, 'numpy')
        c = list(coords) + [0.0] * (3 - len(coords))
        return np.asarray(f(c[0], c[1], c[2], t), dtype=np.float64)

    def is_zero(self):
        return bool(self.expression.is_zero)


def grid_coords(cfg):
    """x_i = i*L/n per axis, arrays shaped for broadcasting over [i,j,k] (DMDA uniform periodic coordinates)."""
    out = []
    for a in range(cfg.dim):
        sh = [1] * cfg.dim
        sh[a] = cfg.n[a]
        out.append((np.arange(cfg.n[a]) * (cfg.L[a] / cfg.n[a])).reshape(sh))
    return out


def decode_sources(sargs, params):
    """ksfdsolver2.py:473-498: --source=<field>=<expr> per field; missing ones are 0."""
    names = params.field_names()
    srcs = [SpatialExpression(params, '0.0') for _ in names]
    seen = set()
    for s in sargs:
        k, val = s.split('=', 1)
        if k in seen:
            raise ValueError('duplicated sources: ' + k)
        if k not in names:
            raise ValueError('unknown function: ' + k)
        seen.add(k)
        srcs[names.index(k)] = SpatialExpression(params, val)
    return srcs


def step_opts_from(params, petsc_args):
    """rtol/atol (ksfdsolver2.py:720-721) + the TSAdapt flags of the --petsc block (options84:47-67)."""
    v = params.values0
    o = default_step_opts(rtol=float(v['rtol']), atol=float(v['atol']))
    a = list(petsc_args)
    i = 0
    while i < len(a):
        k = a[i]
        nxt = a[i + 1] if i + 1 < len(a) and not a[i + 1].startswith('-') or (i + 1 < len(a) and _isnum(a[i + 1])) else None
        if k == '-ts_type' and nxt not in (None, 'rosw'):
            raise ValueError('only -ts_type rosw (ra34pw2) is implemented by the HIP stepper, got %s' % nxt)
        elif k == '-ts_adapt_type':
            o.adapt = 0 if nxt == 'none' else 1
        elif k == '-ts_adapt_clip' and nxt:
            lo, hi = nxt.split(',')
            o.clip_lo, o.clip_hi = float(lo), float(hi)
        elif k == '-ts_adapt_dt_max' and nxt:
            o.dt_max = float(nxt)
        elif k == '-ts_adapt_dt_min' and nxt:
            o.dt_min = float(nxt)
        elif k == '-ts_max_reject' and nxt:
            o.max_reject = int(nxt)
        elif k == '-ksp_rtol' and nxt:
            o.ksp_rtol = float(nxt)
        elif k == '-ksp_atol' and nxt:
            o.ksp_atol = float(nxt)
        elif k == '-ksp_gmres_restart' and nxt:
            o.ksp_restart = int(nxt)
        elif k == '-ksp_max_it' and nxt:
            o.ksp_max_it = int(nxt)
        i += 2 if nxt is not None else 1
    return o


def _isnum(s):
    try:
        float(s.split(',')[0])
        return True
    except ValueError:
        return False
