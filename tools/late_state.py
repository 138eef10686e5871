"""Save the state of the options81-style 384^2 run at a late time (h ~ 300, the slowly converging phase) for solver experiments."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from ksfd_amd import lib as klib
from ksfd_amd.config import ProblemConfig
from ksfd_amd.initial import start_values
n, nsteps, out = 384, int(sys.argv[1]), sys.argv[2]
cfg = ProblemConfig.standard(2, (n, n), L=(1.0, 1.0), nlig=2)
ks = klib.KSFDHip(cfg)
ks.set_state(start_values(cfg))
opts = klib.default_step_opts(adapt=1, atol=0.01, rtol=1e-6)
t, h = 0.0, 1e-8
for s in range(nsteps):
    t, h, st, rc = ks.step(t, h, opts)
np.savez(out, u=ks.get_state(), t=t, h=h)
print('saved', out, 't', t, 'h', h)
