"""Spectral preconditioner: time per application (three kernels) and per-kernel split via the profile, 4096^2 by default."""
import sys
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from bench import build_problem
from ksfd_amd import lib as klib
from ksfd_amd.initial import start_values
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
nlig = int(sys.argv[2]) if len(sys.argv) > 2 else 1
cfg = build_problem(n, nlig)
ks = klib.KSFDHip(cfg)
ks.set_state(start_values(cfg))
ms, by = ks.bench_kernel(klib.KC_SPECTRAL, 20)
print('n=%d nlig=%d spectral apply %.4f ms  (%.1f GB/s of implementation bytes %.0f MB)' % (n, nlig, ms, by / ms / 1e6, by / 1e6), flush=True)
ms, by = ks.bench_kernel(klib.KC_JVP, 20)
print('jvp %.4f ms' % ms)
