"""Rows per wave segment of the strip kernels vs time (frozen Jacobian action, mode 1, and RHS) on the n^2 state."""
import sys
sys.path.insert(0, '.')
from bench import build_problem
from ksfd_amd import lib as klib
from ksfd_amd.initial import reference_rng
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cfg = build_problem(n, 1)
ks = klib.KSFDHip(cfg)
ks.set_state_random(reference_rng().normal(size=(n // 4, n // 4)) * 90.0, 9000.0)
t, h, st, rc = ks.step(0.0, 0.01)                        # frozen coefficient planes exist
for yseg in (8, 12, 14, 15, 16, 17, 18, 20, 24, 31, 32, 48, 62, 64):
    ks.set_tuning(yseg=yseg, yseg_jvp=yseg)
    r = []
    for cls in (klib.KC_JVP, klib.KC_RHS):
        ms, by = ks.bench_kernel(cls, 30)
        r.append('%.4f ms %6.0f GB/s' % (ms, by / ms / 1e6))
    nseg = (n + yseg - 1) // yseg
    waves = ((n + 123) // 124) * nseg
    print('yseg %3d waves %6d (%.2f x 2048)  jvp %s   rhs %s' % (yseg, waves, waves / 2048.0, r[0], r[1]), flush=True)
