"""CPU: pin the oracle (oracle/ksfd_oracle.c) against the golden vectors generated from the
reference's own symbolic + generated-C operator layer (tests/golden/make_golden.py)."""
import numpy as np
import pytest

from conftest import golden_cases, load_golden, rel_l2
from ksfd_amd.config import ProblemConfig
from ksfd_amd.layout import cijk_to_soa
from oracle import ko

OP_TOL = 1e-13        # rel-L2, SURVEY.md 8c


@pytest.mark.parametrize('name', golden_cases('op_'))
def test_oracle_operators(name):
    z = load_golden(name)
    cfg = ProblemConfig.from_golden(z)
    o = ko.Oracle(cfg)
    u = cijk_to_soa(z['u'])
    assert rel_l2(o.rhs(u), cijk_to_soa(z['rhs'])) < OP_TOL
    assert rel_l2(o.velocity(u), cijk_to_soa(z['vel'])) < OP_TOL
    assert rel_l2(o.jvp(u, cijk_to_soa(z['v'])), cijk_to_soa(z['Jv'])) < OP_TOL


@pytest.mark.parametrize('name', golden_cases('op_'))
def test_oracle_groom_active(name):
    """inputs with negatives, sub-floor values and NaNs (KSFD/ksfdsym.py:888-900)"""
    z = load_golden(name)
    o = ko.Oracle(ProblemConfig.from_golden(z))
    ug = cijk_to_soa(z['ug'])
    assert np.isnan(ug).any() and (ug < 0).any()
    r = o.rhs(ug)
    assert np.isfinite(r).all()
    assert rel_l2(r, cijk_to_soa(z['rhs_g'])) < OP_TOL
    assert rel_l2(o.velocity(ug), cijk_to_soa(z['vel_g'])) < OP_TOL
