"""TimeSeries writer fed from the device state (row (f)2 of SURVEY.md section 8: "next", not hot path).

Schema follows KSFD/ksfdtimeseries.py (file <prefix>s<size>r<rank>.h5; /grid/* attributes :253-262;
data<k> float64 (dof,nx,ny[,nz]) C order with attrs k,t :484-509; /times, /ks; /info/dt, /info/lastvart).
h5py is NOT installed in this image, so the HDF5 branch below could not be exercised here; without h5py the
same logical content goes to <prefix>s<size>r<rank>.npz (keys data<k>, t<k>, times, ks, grid_*, info_*), which
is what the tests read back.  The /info dill blobs of the reference (ksfdtsmaker.py:10-29) are not written.
"""
import os

import numpy as np

from .layout import HDF5

try:                                    # pragma: no cover - absent in this image
    import h5py
except ImportError:                     # noqa: D401
    h5py = None


class TimeSeries:
    def __init__(self, basename, grid, mode='w', comm=None, backend=None):
        self.grid = grid
        self.rank = getattr(getattr(grid, 'comm', None), 'rank', 0)
        self.size = getattr(getattr(grid, 'comm', None), 'size', 1)
        self.backend = backend or ('h5' if h5py is not None else 'npz')
        self.filename = '%ss%dr%d.%s' % (basename, self.size, self.rank, self.backend)
        d = os.path.dirname(os.path.abspath(self.filename))
        os.makedirs(d, exist_ok=True)
        self.ks, self.ts = [], []
        self.info = {}
        self.lastk = -1
        self._data = {}
        self.tsFile = None
        if self.backend == 'h5':        # pragma: no cover
            self.tsFile = h5py.File(self.filename, mode)
            g = self.tsFile.require_group('grid')
            for a in ('dim', 'dof', 'nps', 'bounds', 'spacing', 'stencil_width', 'globalSshape', 'globalVshape',
                      'Slshape', 'Vlshape', 'ranges'):
                g[a] = np.asarray(getattr(grid, a))
            self.tsFile.require_group('info')

    def store(self, data, t, k=None):
        """data: Vec-like with .array in PETSc layout, a DeviceVec, or an ndarray in PETSc layout."""
        ks = getattr(data, '_ks', None)
        if ks is not None:
            C = ks.get_state(HDF5).reshape(self.grid.Vlshape)          # device -> (dof,nx,ny[,nz]) C order directly
        else:
            a = np.asarray(getattr(data, 'array', data))
            C = np.ascontiguousarray(a.reshape(self.grid.Vlshape, order='F'))
        if k is None:
            k = self.lastk + 1
        self.lastk = k
        self.ks.append(k)
        self.ts.append(t)
        if self.backend == 'h5':        # pragma: no cover
            ds = self.tsFile.require_dataset('data%d' % k, self.grid.Vlshape, dtype=C.dtype)
            ds.write_direct(C)
            ds.attrs['k'] = k
            ds.attrs['t'] = t
            self.tsFile.flush()
        else:
            self._data['data%d' % k] = C
            self._data['t%d' % k] = np.float64(t)
            self._flush_npz()

    def set_dt(self, h):
        self.info['dt'] = float(h)

    def _flush_npz(self):
        g = self.grid
        meta = {('grid_' + a): np.asarray(getattr(g, a)) for a in
                ('dim', 'dof', 'nps', 'bounds', 'spacing', 'stencil_width', 'globalVshape', 'Vlshape', 'ranges')}
        meta.update({('info_' + k): np.asarray(v) for k, v in self.info.items()})
        tmp = self.filename + '.tmp.npz'
        np.savez(tmp, times=np.array(self.ts), ks=np.array(self.ks), **meta, **self._data)
        os.replace(tmp, self.filename)          # the file is valid after every step (ksfdts.py:481-495 rationale)

    def temp_close(self):
        pass

    def reopen(self):
        pass

    def flush(self):
        if self.backend == 'npz':
            self._flush_npz()

    def close(self):
        if self.backend == 'h5':        # pragma: no cover
            self.tsFile['times'] = np.array(self.ts)
            self.tsFile['ks'] = np.array(self.ks)
            for k, v in self.info.items():
                self.tsFile['info'][k] = v
            self.tsFile.close()
        else:
            self._flush_npz()


def read_last(basename, size=1, rank=0):
    """Last stored point of a series written by this module: (k, t, data[(dof,nx,ny[,nz])], info dict).
    Counterpart of what resume_values reads through KSFD.TimeSeries (ksfdsolver2.py:525-578)."""
    fn = '%ss%dr%d.npz' % (basename, size, rank)
    if not os.path.exists(fn):
        raise FileNotFoundError(fn + ' (only the .npz backend can be resumed from in this image: no h5py)')
    z = np.load(fn)
    ks, ts = z['ks'], z['times']
    i = int(np.argmax(ts))
    info = {k[5:]: z[k][()] for k in z.files if k.startswith('info_')}
    return int(ks[i]), float(ts[i]), z['data%d' % int(ks[i])], info, (ts, ks)
