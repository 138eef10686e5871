"""Minimal HDF5 writer/reader on top of libhdf5's C API through ctypes.

h5py is not installed in this image, but libhdf5 (1.10.x, serial) ships with it; this is just enough of the
C API to write and read back the reference's TimeSeries schema (KSFD/ksfdtimeseries.py): groups, float64/int64
datasets of any rank (C order), scalar datasets, float64/int64 attributes, link deletion, flush.
"""
import ctypes as C
import ctypes.util
import os

import numpy as np

_L = None
H5P_DEFAULT, H5S_ALL = 0, 0
H5F_ACC_RDONLY, H5F_ACC_RDWR, H5F_ACC_TRUNC = 0, 1, 2
H5S_SCALAR = 0
H5F_SCOPE_GLOBAL = 1
hid_t = C.c_int64


def find_lib():
    cands = [os.environ.get('KSFD_HDF5_LIB'), ctypes.util.find_library('hdf5'),
             '/opt/conda/lib/libhdf5.so', '/usr/lib/x86_64-linux-gnu/hdf5/serial/libhdf5.so']
    for c in cands:
        if c and (os.path.exists(c) or '/' not in c):
            try:
                return C.CDLL(c)
            except OSError:
                continue
    return None


def lib():
    global _L
    if _L is None:
        L = find_lib()
        if L is None:
            raise ImportError('libhdf5 not found (set KSFD_HDF5_LIB)')
        L.H5open()
        for name, res, args in [
            ('H5Fcreate', hid_t, [C.c_char_p, C.c_uint, hid_t, hid_t]), ('H5Fopen', hid_t, [C.c_char_p, C.c_uint, hid_t]),
            ('H5Fclose', C.c_int, [hid_t]), ('H5Fflush', C.c_int, [hid_t, C.c_int]),
            ('H5Gcreate2', hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t]), ('H5Gopen2', hid_t, [hid_t, C.c_char_p, hid_t]),
            ('H5Gclose', C.c_int, [hid_t]),
            ('H5Screate_simple', hid_t, [C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
            ('H5Screate', hid_t, [C.c_int]), ('H5Sclose', C.c_int, [hid_t]),
            ('H5Sget_simple_extent_ndims', C.c_int, [hid_t]),
            ('H5Sget_simple_extent_dims', C.c_int, [hid_t, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
            ('H5Dcreate2', hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t, hid_t]),
            ('H5Dopen2', hid_t, [hid_t, C.c_char_p, hid_t]), ('H5Dclose', C.c_int, [hid_t]),
            ('H5Dwrite', C.c_int, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
            ('H5Dread', C.c_int, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
            ('H5Dget_space', hid_t, [hid_t]), ('H5Dget_type', hid_t, [hid_t]),
            ('H5Tget_class', C.c_int, [hid_t]), ('H5Tclose', C.c_int, [hid_t]),
            ('H5Acreate2', hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t]),
            ('H5Aopen', hid_t, [hid_t, C.c_char_p, hid_t]), ('H5Awrite', C.c_int, [hid_t, hid_t, C.c_void_p]),
            ('H5Aread', C.c_int, [hid_t, hid_t, C.c_void_p]), ('H5Aclose', C.c_int, [hid_t]),
            ('H5Aexists', C.c_int, [hid_t, C.c_char_p]), ('H5Adelete', C.c_int, [hid_t, C.c_char_p]),
            ('H5Lexists', C.c_int, [hid_t, C.c_char_p, hid_t]), ('H5Ldelete', C.c_int, [hid_t, C.c_char_p, hid_t]),
            ('H5Eset_auto2', C.c_int, [hid_t, C.c_void_p, C.c_void_p]),
        ]:
            f = getattr(L, name)
            f.restype, f.argtypes = res, args
        L.H5Eset_auto2(0, None, None)                       # errors come back as return codes
        L.T_DOUBLE = hid_t.in_dll(L, 'H5T_NATIVE_DOUBLE_g').value
        L.T_INT64 = hid_t.in_dll(L, 'H5T_NATIVE_INT64_g').value
        _L = L
    return _L


def _chk(x, what):
    if x < 0:
        raise OSError('HDF5 call failed: ' + what)
    return x


def _np_and_type(val):
    a = np.asarray(val)
    # np.array(..., order='C'), not np.ascontiguousarray: the latter turns a 0-d value into shape (1,), and the
    # reference's reader tells scalars from arrays by np.isscalar (KSFD/ksfdtimeseries.py:264-291)
    if a.dtype.kind in 'iub':
        return np.array(a, dtype=np.int64, order='C'), lib().T_INT64
    return np.array(a, dtype=np.float64, order='C'), lib().T_DOUBLE


class File:
    def __init__(self, name, mode='w'):
        L = lib()
        self.name = name
        if mode in ('w', 'x') or (mode in ('a', 'r+') and not os.path.exists(name)):
            self.id = _chk(L.H5Fcreate(name.encode(), H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT), 'H5Fcreate ' + name)
        else:
            self.id = _chk(L.H5Fopen(name.encode(), H5F_ACC_RDONLY if mode == 'r' else H5F_ACC_RDWR, H5P_DEFAULT), 'H5Fopen ' + name)

    def __bool__(self):
        return self.id is not None

    def exists(self, path):
        parts = [p for p in path.split('/') if p]
        cur = ''
        for p in parts:
            cur += '/' + p
            if lib().H5Lexists(self.id, cur.encode(), H5P_DEFAULT) <= 0:
                return False
        return True

    def require_group(self, path):
        L = lib()
        cur = ''
        for p in [q for q in path.split('/') if q]:
            cur += '/' + p
            if L.H5Lexists(self.id, cur.encode(), H5P_DEFAULT) > 0:
                continue
            g = _chk(L.H5Gcreate2(self.id, cur.encode(), H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT), 'H5Gcreate2 ' + cur)
            L.H5Gclose(g)

    def delete(self, path):
        if self.exists(path):
            _chk(lib().H5Ldelete(self.id, path.encode(), H5P_DEFAULT), 'H5Ldelete ' + path)

    def write(self, path, val, attrs=None):
        """(Re)create dataset `path` with the value (scalar or array, C order); optional dict of scalar attributes."""
        L = lib()
        a, ty = _np_and_type(val)
        parent = path.rsplit('/', 1)[0]
        if parent:
            self.require_group(parent)
        self.delete(path)
        if a.ndim == 0:
            sp = _chk(L.H5Screate(H5S_SCALAR), 'H5Screate')
        else:
            dims = (C.c_uint64 * a.ndim)(*a.shape)
            sp = _chk(L.H5Screate_simple(a.ndim, dims, None), 'H5Screate_simple')
        ds = _chk(L.H5Dcreate2(self.id, path.encode(), ty, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT), 'H5Dcreate2 ' + path)
        if a.size:
            _chk(L.H5Dwrite(ds, ty, H5S_ALL, H5S_ALL, H5P_DEFAULT, a.ctypes.data_as(C.c_void_p)), 'H5Dwrite ' + path)
        for k, v in (attrs or {}).items():
            av, aty = _np_and_type(v)
            asp = _chk(L.H5Screate(H5S_SCALAR), 'H5Screate')
            at = _chk(L.H5Acreate2(ds, k.encode(), aty, asp, H5P_DEFAULT, H5P_DEFAULT), 'H5Acreate2 ' + k)
            _chk(L.H5Awrite(at, aty, av.ctypes.data_as(C.c_void_p)), 'H5Awrite ' + k)
            L.H5Aclose(at)
            L.H5Sclose(asp)
        L.H5Dclose(ds)
        L.H5Sclose(sp)

    def read(self, path, attrs=()):
        L = lib()
        ds = _chk(L.H5Dopen2(self.id, path.encode(), H5P_DEFAULT), 'H5Dopen2 ' + path)
        sp = L.H5Dget_space(ds)
        nd = L.H5Sget_simple_extent_ndims(sp)
        dims = (C.c_uint64 * max(nd, 1))()
        if nd > 0:
            L.H5Sget_simple_extent_dims(sp, dims, None)
        shape = tuple(int(dims[i]) for i in range(nd))
        ty = L.H5Dget_type(ds)
        is_int = L.H5Tget_class(ty) == 0                     # H5T_INTEGER
        L.H5Tclose(ty)
        out = np.empty(shape, dtype=np.int64 if is_int else np.float64)
        if out.size:
            _chk(L.H5Dread(ds, L.T_INT64 if is_int else L.T_DOUBLE, H5S_ALL, H5S_ALL, H5P_DEFAULT, out.ctypes.data_as(C.c_void_p)), 'H5Dread ' + path)
        av = {}
        for k in attrs:
            at = _chk(L.H5Aopen(ds, k.encode(), H5P_DEFAULT), 'H5Aopen ' + k)
            v = np.empty((), dtype=np.float64)
            L.H5Aread(at, L.T_DOUBLE, v.ctypes.data_as(C.c_void_p))
            av[k] = float(v)
            L.H5Aclose(at)
        L.H5Sclose(sp)
        L.H5Dclose(ds)
        val = out if nd else out[()]
        return (val, av) if attrs else val

    def flush(self):
        lib().H5Fflush(self.id, H5F_SCOPE_GLOBAL)

    def close(self):
        if self.id is not None:
            lib().H5Fclose(self.id)
            self.id = None
