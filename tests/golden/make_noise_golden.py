#!/usr/bin/env python3
"""Golden vectors for the noise injection of the outer loop, produced by the REFERENCE's own methods
KSFDTS.count_worms / add_variance / conserve_worms / is_noise_time (KSFD/ksfdts.py:239-284), called unbound on a data-only
stand-in for `self` (time, parameter dictionary, local shape, a size-1 communicator) -- container only.
Writes tests/golden/noise_2d.npz.      usage: cd /tmp && python /root/repo/tests/golden/make_noise_golden.py
"""
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(HERE, '_stubs'), '/root/reference']

import numpy as np                                   # noqa: E402
import KSFD                                          # noqa: E402,F401
from KSFD.ksfdts import KSFDTS                       # noqa: E402
from KSFD.ksfdrandom import Generator                # noqa: E402


class Vec:
    def __init__(self, a):
        self.array = a

    def assemble(self):
        pass


def main():
    F, nx, ny = 3, 10, 8
    seed, vrate, interval, dt, t = 5, 2.5e-4, 0.01, 0.012, 0.037
    rng0 = np.random.default_rng(11)
    u0 = 9000.0 + 90.0 * rng0.standard_normal(F * nx * ny)            # local Vec, dof fastest
    params = lambda tt: {'variance_rate': vrate, 'variance_timing_function': tt / interval}
    me = types.SimpleNamespace(
        mpi_comm=types.SimpleNamespace(allreduce=lambda x, op=None: x, rank=0, size=1),
        derivs=types.SimpleNamespace(grid=types.SimpleNamespace(Vlshape=(F, nx, ny)), ps=types.SimpleNamespace(values=params)),
        getTime=lambda: t)
    Generator(seed=seed)                                               # the process-wide stream add_variance draws from
    u = Vec(u0.copy())
    N0 = KSFDTS.count_worms(me, u)
    KSFDTS.add_variance(me, u, dt)
    u_var = u.array.copy()
    KSFDTS.conserve_worms(me, u, N0)
    u_cons = u.array.copy()
    times = np.array([[0.0, 0.0], [0.0099, 0.0], [0.0100001, 0.0], [0.025, 0.02], [0.0301, 0.02], [5.0, 4.5]])
    fire = np.array([bool(KSFDTS.is_noise_time(me, a, b)) for a, b in times])
    np.savez(os.path.join(HERE, 'noise_2d.npz'), F=F, n=np.array([nx, ny]), seed=seed, vrate=vrate, interval=interval, dt=dt, t=t,
             u0=u0, N0=N0, u_var=u_var, u_cons=u_cons, times=times, fire=fire)
    print('N0 %.6f  max |log ratio| %.3e  fire %s' % (N0, np.abs(np.log(u_var[0::F] / u0[0::F])).max(), fire.astype(int)))


if __name__ == '__main__':
    main()
