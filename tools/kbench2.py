import sys
sys.path.insert(0, '.')
from bench import build_problem
from ksfd_amd import lib as klib
from ksfd_amd.initial import start_values
cfg = build_problem(4096, 1)
ks = klib.KSFDHip(cfg)
ks.set_state(start_values(cfg))
for rep in range(2):
    for yseg in (12, 14, 16, 17, 18, 20, 24, 30, 35, 48):
        ks.set_tuning(yseg=yseg, yseg_jvp=yseg)
        ms, by = ks.bench_kernel(klib.KC_JVP, 30)
        ms2, by2 = ks.bench_kernel(klib.KC_RHS, 10)
        print('yseg %3d jvp %.4f ms %.0f GB/s | rhs %.4f ms %.0f GB/s' % (yseg, ms, by / ms / 1e6, ms2, by2 / ms2 / 1e6), flush=True)
