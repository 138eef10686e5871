"""Permissive stand-in: the golden generator never touches PETSc objects; the reference
only needs the names to exist at import time (and Vec/Mat as real classes for isinstance)."""
class _Any:
    def __getattr__(self, k): return _Any()
    def __call__(self, *a, **k): return _Any()
class _PETSc(_Any):
    class Vec: pass
    class Mat: pass
    class Comm: pass
    class TS:
        class Type: ROSW = 'rosw'
        class ExactFinalTime: STEPOVER = 0
    # ^ base class of KSFD.ksfdts.KSFDTS (tests/golden/make_noise_golden.py calls its plain-Python methods unbound)
PETSc = _PETSc()
def init(*a, **k): pass
