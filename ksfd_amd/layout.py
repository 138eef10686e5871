"""Array-layout helpers shared by the host side and the tests.

Three layouts of one (F fields x grid) state occur (include/ksfd_hip.h: KSFD_LAYOUT_*):
  PETSC (0): the reference's DMDA Vec -- dof fastest, then x, y, z: c + F*(i + nx*(j + ny*k))
             (KSFD/ksfdgrid.py:9-58); `vec.array.reshape(Vlshape, order='F')[c,i,j,k]`.
  SOA   (1): device order -- one plane per field, x fastest: i + nx*(j + ny*(k + nz*c)).
  HDF5  (2): TimeSeries datasets (dof,nx,ny[,nz]) C-order, last axis fastest
             (KSFD/ksfdtimeseries.py:484-509).
"""
import numpy as np

PETSC, SOA, HDF5 = 0, 1, 2


def cijk_to_soa(a):
    """a[c,i,j,k] (any memory order, 1-3 spatial axes) -> flat SoA, x fastest."""
    a = np.asarray(a)
    return np.concatenate([a[c].ravel(order='F') for c in range(a.shape[0])])


def soa_to_cijk(flat, F, shape):
    flat = np.asarray(flat).reshape(F, -1)
    return np.stack([flat[c].reshape(shape, order='F') for c in range(F)])


def cijk_to_petsc(a):
    return np.asarray(a).ravel(order='F')


def petsc_to_cijk(flat, F, shape):
    return np.asarray(flat).reshape((F,) + tuple(shape), order='F')


def cijk_to_hdf5(a):
    return np.ascontiguousarray(a).ravel()


def hdf5_to_cijk(flat, F, shape):
    return np.asarray(flat).reshape((F,) + tuple(shape))
