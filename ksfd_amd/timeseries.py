"""TimeSeries writer fed from the device state (row (f)2 of SURVEY.md section 8: "next", not hot path).

Schema = KSFD/ksfdtimeseries.py: file <prefix>s<size>r<rank>.h5 (:188-243); root datasets size, rank, ranges, times,
order, ks, lastk (:122-138, :378-391); /grid/<attr> datasets (:253-262); data<k> float64 (dof,nx,ny[,nz]) C order with
attributes k, t (:484-509); /info/dt, /info/lastvart.  The file is flushed after every store so a crash leaves it
valid (the reference closes and reopens it per step for the same reason, ksfdts.py:481-495).
h5py is not installed in this image: the HDF5 backend goes through libhdf5 directly (ksfd_amd/h5lite.py); if libhdf5
is not found either, the same logical content is written to <prefix>s<size>r<rank>.npz.
Not written: the /info dill blobs (commandlineArguments, SolutionParameters, sources; ksfdtsmaker.py:10-29) -- they
pickle the reference's own classes.
"""
import os

import numpy as np

from .layout import HDF5

try:
    from . import h5lite
    h5lite.lib()
    HAVE_H5 = True
except (ImportError, OSError):
    HAVE_H5 = False

GRID_ATTRS = ('dim', 'dof', 'nps', 'bounds', 'spacing', 'order', 'stencil_width', 'stencil_type', 'boundary_type',
              'globalSshape', 'globalVshape', 'Slshape', 'Vlshape', 'ranges')


def _grid_value(grid, a):
    if a == 'order':
        return getattr(grid, a, 3)
    if a == 'stencil_type':
        return getattr(grid, a, 0)           # DMDA STAR
    if a == 'boundary_type':
        return getattr(grid, a, 3)           # DM_BOUNDARY_PERIODIC
    return getattr(grid, a)


class TimeSeries:
    def __init__(self, basename, grid, mode='w', comm=None, backend=None, async_save=False, save_every=1):
        """async_save: device states are copied out on a third HIP stream (ksfd_snapshot_begin/_wait) and written by a
        background thread, so the stepper does not wait for PCIe or the file system; at most two snapshots are in
        flight and a store blocks when both are.  save_every=N keeps every N-th store (decimation; the reference
        stores every step, ksfdts.py:466-497)."""
        self.async_save, self.save_every, self._nstore = bool(async_save), max(1, int(save_every)), 0
        self._pool, self._outstanding = None, []
        self.grid = grid
        self.rank = getattr(getattr(grid, 'comm', None), 'rank', 0)
        self.size = getattr(getattr(grid, 'comm', None), 'size', 1)
        self.backend = backend or ('h5' if HAVE_H5 else 'npz')
        self.filename = '%ss%dr%d.%s' % (basename, self.size, self.rank, self.backend)
        os.makedirs(os.path.dirname(os.path.abspath(self.filename)), exist_ok=True)
        self.ks, self.ts = [], []
        self.info = {}
        self.lastk = -1
        self._data = {}
        self.tsFile = None
        if self.backend == 'h5':
            f = self.tsFile = h5lite.File(self.filename, mode)
            f.require_group('/info')
            f.write('/size', self.size)
            f.write('/rank', self.rank)
            f.write('/ranges', np.asarray(grid.ranges))
            for a in GRID_ATTRS:
                f.write('/grid/' + a, np.asarray(_grid_value(grid, a)))
            f.flush()

    def store(self, data, t, k=None):
        """data: Vec-like with .array in PETSc layout, a DeviceVec, or an ndarray in PETSc layout."""
        ks = getattr(data, '_ks', None)
        if k is None:
            k = self.lastk + 1
        self.lastk = k
        self._nstore += 1
        if (self._nstore - 1) % self.save_every:
            return
        if ks is not None and self.async_save:
            self._drain(1)                                              # the slot about to be reused must have been written
            slot = ks.snapshot_begin(HDF5)
            if self._pool is None:
                from concurrent.futures import ThreadPoolExecutor
                self._pool = ThreadPoolExecutor(max_workers=1)          # one writer: stores stay in order
            self._outstanding.append(self._pool.submit(self._write_snapshot, ks, slot, int(k), float(t)))
            return
        if ks is not None:
            Cv = ks.get_state(HDF5).reshape(self.grid.Vlshape)         # device -> (dof,nx,ny[,nz]) C order directly
        else:
            a = np.asarray(getattr(data, 'array', data))
            Cv = np.ascontiguousarray(a.reshape(self.grid.Vlshape, order='F'))
        self._commit(k, t, Cv)

    def _write_snapshot(self, ks, slot, k, t):
        self._commit(k, t, ks.snapshot_wait(slot).reshape(self.grid.Vlshape))

    def _drain(self, keep=0):
        while len(self._outstanding) > keep:
            self._outstanding.pop(0).result()                           # re-raises a writer exception here

    def _commit(self, k, t, Cv):
        self._lastk_written = int(k)
        self.ks.append(int(k))
        self.ts.append(float(t))
        if self.backend == 'h5':
            self.tsFile.write('/data%d' % k, Cv, attrs={'k': int(k), 't': float(t)})
            self._write_index()
            self.tsFile.flush()
        else:
            self._data['data%d' % k] = np.array(Cv)                     # own copy: Cv may be a view of a pinned staging buffer
            self._data['t%d' % k] = np.float64(t)
            self._flush_npz()

    def _write_index(self):
        f = self.tsFile
        ts = np.array(self.ts)
        f.write('/times', ts)
        f.write('/order', np.argsort(ts))
        f.write('/ks', np.array(self.ks))
        f.write('/lastk', getattr(self, '_lastk_written', self.lastk))     # async: the main thread may be ahead of the file
        for key, v in self.info.items():
            f.write('/info/' + key, v)

    def set_dt(self, h):
        self.info['dt'] = float(h)

    def _flush_npz(self):
        g = self.grid
        meta = {('grid_' + a): np.asarray(_grid_value(g, a)) for a in GRID_ATTRS}
        meta.update({('info_' + k): np.asarray(v) for k, v in self.info.items()})
        tmp = self.filename + '.tmp.npz'
        np.savez(tmp, times=np.array(self.ts), ks=np.array(self.ks), **meta, **self._data)
        os.replace(tmp, self.filename)

    def temp_close(self):
        pass

    def reopen(self):
        pass

    def flush(self):
        self._drain()
        if self.backend == 'npz':
            self._flush_npz()
        elif self.tsFile:
            self._write_index()
            self.tsFile.flush()

    def close(self):
        self._drain()
        if self._pool is not None:
            self._pool.shutdown()
            self._pool = None
        if self.backend == 'h5':
            if self.tsFile:
                self._write_index()
                self.tsFile.close()
                self.tsFile = None
        else:
            self._flush_npz()


def read_last(basename, size=1, rank=0):
    """Last stored point of a series written by this module: (k, t, data[(dof,nx,ny[,nz])], info dict, (times, ks)).
    Counterpart of what resume_values reads through KSFD.TimeSeries (ksfdsolver2.py:525-578)."""
    stem = '%ss%dr%d' % (basename, size, rank)
    if os.path.exists(stem + '.h5') and HAVE_H5:
        f = h5lite.File(stem + '.h5', 'r')
        ts, ks = f.read('/times'), f.read('/ks')
        i = int(np.argmax(ts))
        data = f.read('/data%d' % int(ks[i]))
        info = {k: float(np.asarray(f.read('/info/' + k)).reshape(-1)[0]) for k in ('dt', 'lastvart') if f.exists('/info/' + k)}
        f.close()
        return int(ks[i]), float(ts[i]), data, info, (ts, ks)
    if os.path.exists(stem + '.npz'):
        z = np.load(stem + '.npz')
        ks, ts = z['ks'], z['times']
        i = int(np.argmax(ts))
        info = {k[5:]: z[k][()] for k in z.files if k.startswith('info_')}
        return int(ks[i]), float(ts[i]), z['data%d' % int(ks[i])], info, (ts, ks)
    raise FileNotFoundError(stem + '.h5/.npz')


def read_series(basename, size=1, rank=0):
    """All points of a series written by this module: dict(times, ks, data={k: array})."""
    stem = '%ss%dr%d' % (basename, size, rank)
    if os.path.exists(stem + '.h5') and HAVE_H5:
        f = h5lite.File(stem + '.h5', 'r')
        ts, ks = f.read('/times'), f.read('/ks')
        out = dict(times=ts, ks=ks, data={int(k): f.read('/data%d' % int(k)) for k in ks})
        f.close()
        return out
    z = np.load(stem + '.npz')
    return dict(times=z['times'], ks=z['ks'], data={int(k): z['data%d' % int(k)] for k in z['ks']})
