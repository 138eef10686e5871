"""Per-kernel microbenchmark on the 4096^2 state (HIP events inside ksfd_bench_kernel)."""
import sys, json
sys.path.insert(0, '.')
import numpy as np
from bench import build_problem
from ksfd_amd import lib as klib
from ksfd_amd.initial import start_values
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cfg = build_problem(n, 1)
ks = klib.KSFDHip(cfg)
ks.set_state(start_values(cfg))
names = {klib.KC_RHS: 'rhs', klib.KC_JVP: 'jvp', klib.KC_MULTIDOT: 'multidot8', klib.KC_GSUPDATE: 'gs_update8', klib.KC_LINCOMB: 'lincomb3'}
for yseg in (8, 16, 32, 64):
    ks.set_tuning(yseg=yseg)
    for cls in (klib.KC_RHS, klib.KC_JVP):
        ms, by = ks.bench_kernel(cls, 20)
        print('yseg %3d %-10s %.4f ms  %.1f GB/s' % (yseg, names[cls], ms, by / ms / 1e6), flush=True)
print('--- recompute (non-frozen) JVP'); ks.set_tuning(use_fused=3, yseg=32)
ms, by = ks.bench_kernel(klib.KC_JVP, 20); print('recompute jvp %.4f ms %.1f GB/s' % (ms, by/ms/1e6), flush=True)
ks.set_tuning(use_fused=0)
for cls in (klib.KC_RHS, klib.KC_JVP):
    ms, by = ks.bench_kernel(cls, 10)
    print('generic  %-10s %.4f ms  %.1f GB/s (incl. gfield pass)' % (names[cls], ms, by / ms / 1e6), flush=True)
for cls in (klib.KC_MULTIDOT, klib.KC_GSUPDATE, klib.KC_LINCOMB):
    ms, by = ks.bench_kernel(cls, 20)
    print('         %-10s %.4f ms  %.1f GB/s' % (names[cls], ms, by / ms / 1e6), flush=True)
