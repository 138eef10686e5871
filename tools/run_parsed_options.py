"""Run one of the reference's shipped option sets end to end on the GPU WITHOUT the reference's files: the parameter table the
reference's own parser produced for it is a committed fixture (tests/golden/options_parsed.json), and every entry is passed
back as an explicit key=value argument to ksfd_amd.solver.main.
usage: python tools/run_parsed_options.py options81 [--save prefix] [--async_save] [--saveevery N] [--maxsteps N]"""
import json, os, sys, time
sys.path.insert(0, '.')
from ksfd_amd import solver

name = sys.argv[1]
extra = sys.argv[2:]
tab = json.load(open(os.path.join('tests', 'golden', 'options_parsed.json')))[name]
args = []
for k, v in tab['values0'].items():
    if k in ('t', 'lastvart') or v == '' or v is None or isinstance(v, dict):
        continue
    if k == 'maxsteps' and any(a.startswith('--maxsteps') for a in extra):
        continue
    args.append('%s=%r' % (k, v))
if '--maxsteps' in extra:
    i = extra.index('--maxsteps')
    args.append('maxsteps=%d' % int(extra[i + 1]))
    del extra[i:i + 2]
argv = ['ksfd', '--cappotential=' + tab['cappotential'], '--seed=%d' % tab['seed']] + extra + args + ['--petsc'] + tab['petsc'] + ['--']
quiet = os.environ.get('KSFD_QUIET')
if quiet:                                  # drop the per-step print monitor output
    sys.stdout = open(os.devnull, 'w')
t0 = time.perf_counter()
ts = solver.main(*argv)
wall = time.perf_counter() - t0
sys.stdout = sys.__stdout__
print('%s: %d steps to t = %.6g in %.1f s wall (%.2f ms/step), SNES failures %d' % (name, ts.getStepNumber(), ts.getTime(), wall, 1e3 * wall / max(ts.getStepNumber(), 1), ts.getSNESFailures()))
ts.cleanup()
