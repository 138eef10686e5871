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
    assert ctypes.sizeof(klib.StepStats) == 24 + 32
    assert ctypes.sizeof(klib.Profile) == 13 * 8 * 3


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from ksfd_amd.config import ProblemConfig
    with pytest.raises(klib.KSFDError):
        klib.KSFDHip(ProblemConfig.standard(2, (16, 16)))
