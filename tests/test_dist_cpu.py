"""CPU, world_size 2 and 3, gloo: the N>1 host logic -- slab ranges, ring neighbours, the transport-2
callbacks (called through their C function pointers exactly as libksfd_hip.so calls them), and the halo
protocol end to end: every rank fills the 2 ghost rows per side of its slab through HostRing, applies the
ORACLE stencil to its padded slab, and the gathered result must equal the global operator."""
import ctypes as C
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ksfd_amd.config import ProblemConfig
from ksfd_amd.dist import HostRing, gather_slabs, local_slab, neighbours, slab_range


def test_slab_ranges_and_neighbours():
    assert slab_range(4096, 3, 8) == (1536, 2048)
    assert [slab_range(16, r, 4) for r in range(4)] == [(0, 4), (4, 8), (8, 12), (12, 16)]
    with pytest.raises(ValueError):
        slab_range(10, 0, 4)          # not divisible
    with pytest.raises(ValueError):
        slab_range(8, 0, 4)           # fewer than 4 rows per rank
    assert neighbours(0, 4) == (3, 1) and neighbours(3, 4) == (2, 0) and neighbours(0, 2) == (1, 1)
    cfg = ProblemConfig.standard(2, (6, 8))
    u = np.arange(2 * 48, dtype=np.float64)
    parts = [local_slab(u, cfg, r, 2) for r in range(2)]
    assert np.array_equal(parts[0].reshape(2, 4, 6), u.reshape(2, 8, 6)[:, :4])
    assert np.array_equal(parts[1].reshape(2, 4, 6), u.reshape(2, 8, 6)[:, 4:])


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, size, port, shape, nlig):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=size)
    try:
        from oracle import ko
        ring = HostRing()
        d = ring.cdist()
        assert (d.rank, d.size, d.transport) == (rank, size, 2)
        # --- callbacks through their C pointers
        n = 5
        bufs = [np.full(n, 10.0 * rank + k) for k in range(2)] + [np.zeros(n), np.zeros(n)]
        ptr = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
        assert d.exchange(None, ptr(bufs[0]), ptr(bufs[1]), ptr(bufs[2]), ptr(bufs[3]), n) == 0
        lo, hi = neighbours(rank, size)
        assert np.all(bufs[2] == 10.0 * lo + 1)      # recv_lo = lower neighbour's send_hi
        assert np.all(bufs[3] == 10.0 * hi + 0)      # recv_hi = upper neighbour's send_lo
        red = np.array([rank + 1.0, -rank])
        assert d.allreduce(None, ptr(red), 2, 0) == 0
        assert red[0] == size * (size + 1) / 2
        mx = np.array([float(rank)])
        assert d.allreduce(None, ptr(mx), 1, 1) == 0 and mx[0] == size - 1
        # all-to-all (transposes of the slab-distributed spectral solver): block q of `send` -> rank q
        nb = 24
        send = np.concatenate([np.full(nb, 16 * rank + q, dtype=np.uint8) for q in range(size)])
        recv = np.zeros(nb * size, dtype=np.uint8)
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        assert d.alltoall(None, vp(send), vp(recv), nb) == 0
        for r in range(size):
            assert np.all(recv[r * nb:(r + 1) * nb] == 16 * r + rank)       # block r came from rank r and was addressed to us
        # --- halo protocol end to end with the oracle as the stencil
        dim = len(shape)
        cfg = ProblemConfig.standard(dim, shape, L=[0.5 + 0.1 * a for a in range(dim)], nlig=nlig)
        F, N = cfg.F, cfg.N
        rng = np.random.default_rng(7)
        u = 9000 + 900 * rng.standard_normal(F * N)
        s0, s1 = slab_range(cfg.n[dim - 1], rank, size)
        sloc = s1 - s0
        inner = N // cfg.n[dim - 1]
        mine = local_slab(u, cfg, rank, size).reshape(F, sloc, inner)
        padded = np.zeros((F, sloc + 4, inner))
        padded[:, 2:-2] = mine
        slo = np.ascontiguousarray(mine[:, :2]).reshape(-1)
        shi = np.ascontiguousarray(mine[:, -2:]).reshape(-1)
        rlo, rhi = np.empty_like(slo), np.empty_like(shi)
        ring.exchange_arrays(slo, shi, rlo, rhi)
        padded[:, :2] = rlo.reshape(F, 2, inner)
        padded[:, -2:] = rhi.reshape(F, 2, inner)
        nloc = list(cfg.n)
        nloc[dim - 1] = sloc + 4
        Lloc = list(cfg.L)
        Lloc[dim - 1] = cfg.L[dim - 1] * (sloc + 4) / cfg.n[dim - 1]       # same spacing
        cloc = ProblemConfig.standard(dim, nloc[:dim], L=Lloc[:dim], nlig=nlig)
        r_loc = ko.Oracle(cloc).rhs(padded.reshape(-1)).reshape(F, sloc + 4, inner)[:, 2:-2]
        glob = gather_slabs(np.ascontiguousarray(r_loc).reshape(-1), cfg)
        want = ko.Oracle(cfg).rhs(u)
        err = np.linalg.norm(glob - want) / np.linalg.norm(want)
        assert err < 1e-13, err
        assert not ring.errors
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('size,shape,nlig', [(2, (12, 16), 1), (3, (10, 24), 2), (2, (6, 5, 8), 1)])
def test_halo_protocol_gloo(size, shape, nlig):
    mp.spawn(_worker, args=(size, _free_port(), shape, nlig), nprocs=size, join=True)
