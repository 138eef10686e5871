"""CPU: the C-ABI library loads and exports every symbol include/ksfd_hip.h declares."""
import ctypes
import os
import re

import pytest

from conftest import ROOT
from ksfd_amd import lib as klib


def _declared():
    txt = open(os.path.join(ROOT, 'include', 'ksfd_hip.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(ksfd_[a-z_]+)\s*\(', txt)) - {'ksfd_exchange_fn', 'ksfd_allreduce_fn'})


def test_header_and_binding_agree():
    assert _declared() == sorted(klib.ABI_SYMBOLS)


def test_library_exports_every_symbol():
    if not os.path.exists(klib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    L = ctypes.CDLL(klib.LIB_PATH)
    for s in _declared():
        assert hasattr(L, s), s


def test_struct_sizes_match_header():
    # ksfd_config: 4*int32 + 3*int64 + 3*double + 6*double + 7 pointers
    assert ctypes.sizeof(klib.CConfig) == 16 + 24 + 24 + 48 + 56
    assert ctypes.sizeof(klib.StepOpts) == 8 * 2 + 8 + 8 * 8 + 16
    assert ctypes.sizeof(klib.StepStats) == 24 + 32 + 16
    assert ctypes.sizeof(klib.Profile) == klib.NKCLASS * 8 * 4 and klib.NKCLASS == 14


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from ksfd_amd.config import ProblemConfig
    with pytest.raises(klib.KSFDError):
        klib.KSFDHip(ProblemConfig.standard(2, (16, 16)))


def test_header_is_plain_c_and_a_c_program_links(tmp_path):
    """the boundary is a C ABI: include/ksfd_hip.h must compile as pedantic C99, and a C program linked against
    libksfd_hip.so must be able to call it (no compute without a GPU: defaults, names, the error path of ksfd_create)"""
    import shutil
    import subprocess
    if not shutil.which('gcc'):
        pytest.skip('no gcc')
    if not os.path.exists(klib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    inc = os.path.join(ROOT, 'include')
    subprocess.run(['gcc', '-std=c99', '-Wall', '-Wextra', '-pedantic', '-Werror', '-fsyntax-only', '-x', 'c', os.path.join(inc, 'ksfd_hip.h')], check=True)
    src = tmp_path / 'use_abi.c'
    src.write_text('''
#include "ksfd_hip.h"
#include <stdio.h>
int main(void) {
    ksfd_step_opts o;
    ksfd_config c = {0};
    ksfd_handle *h = 0;
    int rc;
    ksfd_default_step_opts(&o);
    printf("class1=%s rtol=%g pc=%d\\n", ksfd_kernel_class_name(1), o.ksp_rtol, (int)o.pc_type);
    rc = ksfd_create(&c, 0, &h);                 /* dim = 0: rejected before any device is touched */
    printf("rc=%d err=%s\\n", rc, ksfd_last_error(0));
    return rc == 0;
}
''')
    exe = tmp_path / 'use_abi'
    libdir = os.path.dirname(klib.LIB_PATH)
    subprocess.run(['gcc', '-std=c99', '-Wall', '-I', inc, str(src), '-o', str(exe), '-L', libdir, '-lksfd_hip', '-Wl,-rpath,' + libdir], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    assert 'class1=jvp rtol=1e-06 pc=2' in r.stdout and 'rc=1' in r.stdout and 'dim must be' in r.stdout


def test_live_pmc_of_the_bench_never_raises_without_a_gpu():
    """bench.py measures roofline.traffic in two rocprofv3 --pmc child runs of itself; whatever goes wrong there (no profiler, no GPU, a
    crash of the child) must come back as (None, reason) -- the bench line then falls back to the committed summary and says so."""
    import argparse
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
    import bench
    args = argparse.Namespace(n=64, nlig=1, dim=2)
    val, err = bench.live_pmc(args, 'spectral')
    assert val is None and isinstance(err, str) and err
    val, err = bench.live_pmc(args, 'no-such-class')
    assert val is None and 'pattern' in err
