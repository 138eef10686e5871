import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from conftest import load_golden, rel_l2
from ksfd_amd.config import ProblemConfig
from ksfd_amd.layout import cijk_to_soa
from ksfd_amd import lib as klib
from oracle import ko
z = load_golden(sys.argv[1] if len(sys.argv) > 1 else 'step_2d_n1_mild')
cfg = ProblemConfig.from_golden(z)
o = ko.Oracle(cfg)
u0 = cijk_to_soa(z['u0'])
for fused in (1, 0):
    for h in (0.1, 1.0):
        for rt in (1e-12,):
            k = klib.KSFDHip(cfg); k.set_tuning(use_fused=fused)
            k.set_state(u0)
            t, hn, st, rc = k.step(0.0, h, klib.default_step_opts(adapt=0, atol=0.01, rtol=1e-6, ksp_rtol=rt, ksp_max_it=4000), raise_on_error=False)
            un, err, wr, _ = o.rosw_step(u0, h, 0.01, 1e-6, solver='lu')
            print('fused', fused, 'h', h, 'ksp_rtol', rt, 'rc', rc, 'its', st.linear_its, 'resid', st.ksp_resid, 'wrms', st.wrms, 'oracle wrms', wr, 'relL2', rel_l2(k.get_state(), un), k.last_error() if rc else '')
            k.close()
