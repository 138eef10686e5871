#!/usr/bin/env python3
"""Golden vectors for the start-value interpolation, produced by the REFERENCE's own
KSFD.ksfdrandom.random_function (KSFD/ksfdrandom.py:108-220) -- container only, never shipped to the GPU box.

What stands in for PETSc: a grid object that is pure data plumbing (no interpolation arithmetic): point counts,
spacing, coordinate arrays (i*h), a Vec with an `.array`, and a periodic BOX ghost fill (np.pad mode='wrap') for
DMDA.globalToLocal.  The interpolation itself -- KDTree ball query, the weight f(x) = 2x^3 - 3x^2 + 1 per axis, the
product over axes, the accumulation -- is the reference's code, run unmodified.  np.product (removed in numpy 2) is
aliased to np.prod for the duration of the call, as SURVEY.md 8f notes is needed.

Writes tests/golden/randfn_<case>.npz: n (fine points per axis), nc (coarse), L, z (coarse samples indexed [i,j,k]),
out (fine field indexed [i,j,k]).     usage: python tests/golden/make_randfn_golden.py
"""
import importlib.util
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference/KSFD'


class Vec:
    def __init__(self, n):
        self.array = np.zeros(n)

    def assemble(self):
        pass


class FakeDMDA:
    def __init__(self, nps, sw):
        self.nps, self.sw = tuple(int(n) for n in nps), sw

    def createGlobalVec(self):
        return Vec(int(np.prod(self.nps)))

    def createLocalVec(self):
        return Vec(int(np.prod([n + 2 * self.sw for n in self.nps])))

    def globalToLocal(self, g, l):
        a = g.array.reshape(self.nps, order='F')
        l.array = np.pad(a, self.sw, mode='wrap').reshape(-1, order='F')       # periodic, BOX stencil: corners too


class FakeGrid:
    """what random_function / extended_coords read from a KSFD.Grid on one process (KSFD/ksfdgrid.py:140-186, 330-380)"""

    def __init__(self, nps, L, sw=2):
        self.dim = len(nps)
        self.nps = np.array(nps, dtype=int)
        self.bounds = np.array(L, dtype=float)
        self.spacing = self.bounds / self.nps
        self.stencil_width = sw
        self.comm = types.SimpleNamespace(rank=0, size=1)
        self.Slshape = tuple(int(n) for n in nps)
        self.Clshape = (self.dim,) + self.Slshape
        self.Sashape = tuple(int(n) + 2 * sw for n in nps)
        self.Cashape = (self.dim,) + self.Sashape
        self.Sdmda = FakeDMDA(nps, sw)
        axes = [np.arange(n) * h for n, h in zip(self.nps, self.spacing)]
        self.coordsNoGhosts = np.asfortranarray(np.stack(np.meshgrid(*axes, indexing='ij')))
        gaxes = [(np.arange(-sw, n + sw) % n) * h for n, h in zip(self.nps, self.spacing)]   # DMDA ghost coordinates wrap
        self.coordsWithGhosts = np.asfortranarray(np.stack(np.meshgrid(*gaxes, indexing='ij')))


def load_reference_random():
    mpi = types.ModuleType('mpi4py.MPI')
    mpi.COMM_WORLD = types.SimpleNamespace(rank=0, size=1)
    pkg = types.ModuleType('mpi4py')
    pkg.MPI = mpi
    sys.modules['mpi4py'], sys.modules['mpi4py.MPI'] = pkg, mpi
    ks = types.ModuleType('KSFD')
    ks.__path__ = [REF]
    sys.modules['KSFD'] = ks
    for name in ('ksfddebug', 'ksfdrandom'):
        spec = importlib.util.spec_from_file_location('KSFD.' + name, os.path.join(REF, name + '.py'))
        mod = importlib.util.module_from_spec(spec)
        sys.modules['KSFD.' + name] = mod
        spec.loader.exec_module(mod)
    return sys.modules['KSFD.ksfdrandom']


# Only grids with the same point count on every axis (fine and coarse): random_function flattens its coordinate arrays in
# C order (`.reshape(gdim, -1)`, KSFD/ksfdrandom.py:183,193) but reads and writes the PETSc Vecs in x-fastest order; on
# such grids the two index reversals cancel and the result is the tensor-product interpolation, on others it is a
# scramble of it (the reference's own runs use square grids).  The box may be anisotropic.
CASES = {
    '1d': ((24,), (6,), (1.2,)),
    '1d_ragged': ((26,), (5,), (1.0,)),                 # coarse points are not a subset of the fine ones
    '2d': ((16, 16), (4, 4), (0.8, 0.8)),
    '2d_aniso_box_ragged': ((18, 18), (5, 5), (1.0, 0.25)),
    '3d': ((9, 9, 9), (3, 3, 3), (0.4, 0.6, 0.5)),
}


def main():
    R = load_reference_random()
    np.product = np.prod                                 # KSFD/ksfdrandom.py:196 (numpy >= 2 dropped the alias)
    for name, (n, nc, L) in CASES.items():
        rng = np.random.default_rng(np.random.SeedSequence(793817931).spawn(1)[0])     # ksfdsolver2.py:412, ksfdrandom.py:44-49
        z = 9000.0 + 90.0 * rng.normal(size=nc)                                         # ksfdsolver2.py:603-610
        grid, rgrid = FakeGrid(n, L), FakeGrid(nc, L)
        vals = rgrid.Sdmda.createGlobalVec()
        vals.array = z.reshape(-1, order='F').copy()
        out = R.random_function(grid, randgrid=rgrid, vals=vals)
        np.savez(os.path.join(HERE, 'randfn_%s.npz' % name), n=np.array(n), nc=np.array(nc), L=np.array(L), z=z,
                 out=out.array.reshape(n, order='F'))
        print(name, n, nc, 'min %.3f max %.3f' % (out.array.min(), out.array.max()))


if __name__ == '__main__':
    main()
