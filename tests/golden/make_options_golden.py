#!/opt/conda/bin/python3.9
"""Golden tables for the @options front end: every options file the reference ships, parsed by the REFERENCE's own
ksfdsolver2.parse_commandline + KSFD.SolutionParameters (container only; conda python 3.9 because its sympy 1.9 accepts the
blank U0 defaults that sympy 1.14 rejects).  Stand-ins: the arithmetic-free stubs of tests/golden/_stubs for
mpi4py / petsc4py / dogpile, and an empty `dill`.

Writes tests/golden/options_parsed.json:  {file: {values0: {...}, ligands: [[name, group, s, gamma, D, weight], ...],
groups: [[alpha, beta], ...], petsc: [...], cappotential, seed, nsources, save, check}}
(expressions in x, y, z, t are stored as their values at three sample points, never as text)
usage (from any directory):  /opt/conda/bin/python3.9 tests/golden/make_options_golden.py
"""
import json
import os
import sys
import types
import warnings

warnings.filterwarnings('ignore')
HERE = os.path.dirname(os.path.abspath(__file__))
sys.modules.setdefault('dill', types.ModuleType('dill'))
sys.path[:0] = [os.path.join(HERE, '_stubs'), '/root/reference']

import ksfdsolver2 as S                               # noqa: E402
from KSFD import SolutionParameters                   # noqa: E402

FILES = ['options80', 'options81', 'options84', 'options92', 'options93nx128dt1', 'options113a']


SAMPLES = [(0.1, 0.2, 0.3, 0.0), (0.37, 0.11, 0.05, 1.5), (0.9, 0.45, 0.6, 40.0)]       # (x, y, z, t)


def plain(v):
    """numbers as numbers; expressions in (x, y, z, t) as their values at SAMPLES (a printed expression would tie the
    fixture to one sympy version); anything else as text"""
    import sympy as sy
    if isinstance(v, (bool, int)) or v is None:
        return v
    if isinstance(v, str) and v.strip() == '':
        return ''
    try:
        return float(v)
    except (TypeError, ValueError):
        pass
    try:
        e = sy.sympify(v)
        x, y, z, t = sy.symbols('x y z t')
        if e.free_symbols <= {x, y, z, t}:
            return {'samples': [float(e.subs({x: a, y: b, z: c, t: d})) for a, b, c, d in SAMPLES]}
    except Exception:
        pass
    return str(v)


def main():
    out = {}
    for f in FILES:
        path = os.path.join('/root/reference', f)
        if not os.path.exists(path):
            continue
        cl = S.parse_commandline(['@' + path])
        ps = SolutionParameters(cl)
        import sympy as sy
        vals = {sy.Symbol(str(k)): v for k, v in ps.values0.items() if not isinstance(v, (str, bool)) and v is not None}

        def num(e):                                       # ligand / group attributes may be expressions in other parameters
            return float(sy.sympify(e).subs(vals))
        ligs = [[l.name(), int(l.groupnum), num(l.s), num(l.gamma), num(l.D), num(l.weight)] for l in ps.groups.ligands()]
        groups = [[num(g.alpha), num(g.beta)] for g in ps.groups.groups]
        out[f] = dict(values0={str(k): plain(v) for k, v in ps.values0.items()}, ligands=ligs, groups=groups,
                      petsc=list(cl.petsc), cappotential=cl.cappotential, seed=int(cl.seed), nsources=len(cl.source),
                      save=cl.save, check=cl.check)
        print(f, len(out[f]['values0']), 'parameters', len(ligs), 'ligands')
    json.dump(out, open(os.path.join(HERE, 'options_parsed.json'), 'w'), indent=1, sort_keys=True)


if __name__ == '__main__':
    main()
