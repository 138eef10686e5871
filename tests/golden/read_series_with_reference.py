#!/opt/conda/bin/python3.9
"""Container-only checker: open a series written by ksfd_amd.timeseries with the REFERENCE's own reader
(KSFD.ksfdtimeseries.TimeSeries, /root/reference) and dump what it sees.

Runs under the image's conda python 3.9 (the only interpreter here that has h5py).  petsc4py / mpi4py are absent: the
two names are filled with stand-ins that carry no arithmetic -- an MPI communicator of size 1 and a DMDA that only
remembers its sizes and answers getRanges() for one process -- which is all the reader touches (KSFD/ksfdgrid.py:140-186,
KSFD/ksfdtimeseries.py:82-138, 264-314, 551-619).  KSFD/__init__.py is bypassed (it pulls in sympy and the code
generator); ksfddebug, ksfdgrid and ksfdtimeseries are loaded from their files.

usage: read_series_with_reference.py <series prefix> <out.npz> [size rank]
"""
import importlib.util
import os
import sys
import types

import numpy as np

REF = '/root/reference/KSFD'


def _module(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _Comm:
    def __init__(self, rank=0, size=1):
        self.rank, self.size = rank, size

    def bcast(self, x, root=0):
        return x

    def allreduce(self, x, op=None):
        return x

    def Barrier(self):
        pass

    def tompi4py(self):
        return self


class _DMDA:
    class StencilType:
        STAR, BOX = 0, 1

    class BoundaryType:
        NONE, GHOSTED, MIRROR, PERIODIC = 0, 1, 2, 3

    def create(self, dim=None, sizes=None, **kw):
        self.sizes = tuple(int(s) for s in sizes)
        return self

    def getRanges(self):
        return tuple((0, s) for s in self.sizes)

    def __getattr__(self, name):            # setUniformCoordinates, setFromOptions, setUp, ...
        return lambda *a, **k: None


class _Anything:
    def __getattr__(self, k):
        return _Anything()

    def __call__(self, *a, **k):
        return _Anything()


def load_reference_reader(rank=0, size=1):
    petsc = _Anything()
    petsc.DMDA = _DMDA
    petsc.Vec, petsc.Mat, petsc.Comm = type('Vec', (), {}), type('Mat', (), {}), type('Comm', (), {})
    _module('petsc4py', PETSc=petsc, init=lambda *a, **k: None)
    mpi = _module('mpi4py.MPI', COMM_WORLD=_Comm(rank, size), COMM_SELF=_Comm(), INT64_T='int64', SUM='sum', MAX='max')
    _module('mpi4py', MPI=mpi)
    pkg = _module('KSFD')
    pkg.__path__ = [REF]
    for name in ('ksfddebug', 'ksfdgrid', 'ksfdtimeseries'):
        spec = importlib.util.spec_from_file_location('KSFD.' + name, os.path.join(REF, name + '.py'))
        mod = importlib.util.module_from_spec(spec)
        sys.modules['KSFD.' + name] = mod
        spec.loader.exec_module(mod)
    return sys.modules['KSFD.ksfdtimeseries']


def main():
    prefix, out = sys.argv[1], sys.argv[2]
    size, rank = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1, 0)
    T = load_reference_reader(rank, size)
    ts = T.TimeSeries(prefix, comm=_Comm(rank, size), mode='r')
    got = dict(times=np.asarray(ts.times()), steps=np.asarray(ts.steps()), sorted_times=np.asarray(ts.sorted_times()),
               dim=np.int64(ts.dim), dof=np.int64(ts.dof), nps=np.asarray(ts.grid.nps), bounds=np.asarray(ts.grid.bounds),
               Vlshape=np.asarray(ts.grid.Vlshape), ranges=np.asarray(ts.ranges))
    for k in ts.steps():
        got['data%d' % int(k)] = ts.retrieve_by_number(int(k))
    st = np.asarray(ts.sorted_times())
    if st.size >= 2:
        tm = 0.25 * st[0] + 0.75 * st[1]
        got['interp_t'] = np.float64(tm)
        got['interp'] = ts.retrieve_by_time(tm)
    if 'dt' in ts.info:
        got['info_dt'] = np.float64(ts.info['dt'][()])
    ts.close()
    np.savez(out, **got)
    print('reference-reader-ok')


if __name__ == '__main__':
    main()
