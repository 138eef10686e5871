"""GPU: error behaviour at the C ABI (codes + ksfd_last_error message; nothing crashes, nothing falls back)."""
import ctypes as C

import numpy as np
import pytest

from ksfd_amd import lib as klib
from ksfd_amd.config import ProblemConfig

pytestmark = pytest.mark.gpu


def test_create_rejects_bad_configurations():
    with pytest.raises(klib.KSFDError) as e:
        klib.KSFDHip(ProblemConfig.standard(2, (4, 16)))            # the width-2 periodic star needs >= 5 points
    assert e.value.code == klib.EINVAL and '>= 5' in str(e.value)
    d = klib.CDist()
    d.rank, d.size, d.transport, d.device = 0, 3, 2, 0
    with pytest.raises(klib.KSFDError) as e:                        # 16 rows over 3 ranks: not divisible
        klib.KSFDHip(ProblemConfig.standard(2, (16, 16)), d)
    assert e.value.code == klib.EINVAL
    d.size, d.device = 1, 99
    with pytest.raises(klib.KSFDError):
        klib.KSFDHip(ProblemConfig.standard(2, (16, 16)), d)


def test_calls_report_codes_and_messages():
    cfg = ProblemConfig.standard(2, (16, 16))
    k = klib.KSFDHip(cfg)
    u = np.full(k.nlocal, 9000.0)
    with pytest.raises(klib.KSFDError) as e:
        k.set_state(u, layout=7)
    assert e.value.code == klib.EINVAL and 'layout' in str(e.value)
    with pytest.raises(ValueError):
        k.set_state(u[:-1])                                          # wrong length is caught before the C call
    with pytest.raises(klib.KSFDError):
        k.last_error_vector()                                        # no step attempted yet
    with pytest.raises(klib.KSFDError):
        k.update_params(ProblemConfig.standard(2, (32, 16)))         # grid change is not a parameter update
    with pytest.raises(klib.KSFDError):
        k.set_source(5, np.zeros(256))                               # no such field
    assert k.L.ksfd_step(k.h, None, None, None, None) == klib.EINVAL
    assert k.L.ksfd_rhs(None, 0.0, None, None, 0) == klib.EINVAL
    # a state of NaNs is groomed, not an error (Derivatives.groom turns NaN into the floor)
    k.set_state(np.full(k.nlocal, np.nan))
    r = k.rhs()
    assert np.isfinite(r).all()
    # time-dependent parameters: same grid, new physics is fine and changes the operator
    k.set_state(u + np.arange(k.nlocal))
    r1 = k.rhs()
    cfg2 = ProblemConfig.standard(2, (16, 16))
    cfg2.s2 *= 2
    k.update_params(cfg2)
    assert not np.allclose(k.rhs(), r1)
    k.close()
    k.close()                                                        # idempotent
