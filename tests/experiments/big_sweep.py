"""Offline extension of tests/test_gpu_random_sweep.py: many more seeds (not part of the test suite)."""
import os, sys, time, traceback
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import test_gpu_random_sweep as T
nop, nst = int(sys.argv[1]), int(sys.argv[2])
bad = []
t0 = time.time()
for i in range(100, 100 + nop):
    try:
        T.test_random_problem_operators_vs_oracle(i + 5000)
    except Exception as e:
        bad.append(('op', i + 5000, repr(e)[:200]))
print('operators: %d cases, %d failures, %.1f s' % (nop, len(bad), time.time() - t0), flush=True)
t0 = time.time(); nb = len(bad)
for i in range(40, 40 + nst):
    try:
        T.test_random_problem_step_vs_oracle(i + 7000)
    except Exception as e:
        bad.append(('step', i + 7000, repr(e)[:300]))
print('steps: %d cases, %d failures, %.1f s' % (nst, len(bad) - nb, time.time() - t0), flush=True)
for b in bad[:20]:
    print(b)
