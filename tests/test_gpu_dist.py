"""GPU, 2-3 ranks sharing cuda:0: the slab-decomposed path (ghost rows, halo exchange, global reductions,
distributed GMRES/ROSW step) must reproduce the single-rank result.  Transport 2 (host callbacks over gloo) always;
transport 1 (RCCL inside the library) when RCCL accepts several ranks on one device, else skipped."""
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import rel_l2
from ksfd_amd.config import ProblemConfig

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, size, port, shape, nlig, transport, outfile):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=size)
    try:
        from ksfd_amd import lib as klib
        from ksfd_amd.dist import open_handle, local_slab, gather_slabs
        dim = len(shape)
        cfg = ProblemConfig.standard(dim, shape, L=[0.2 + 0.05 * a for a in range(dim)], nlig=nlig)
        rng = np.random.default_rng(3)
        N = cfg.N
        rho = 9000 + 90 * rng.standard_normal(N)
        u = np.concatenate([rho] + [rho * cfg.lig_s[l] / cfg.lig_gamma[l] + rng.standard_normal(N) for l in range(nlig)])
        v = rng.standard_normal(cfg.F * N)
        try:
            ks, keep = open_handle(cfg, rank, size, 0, transport=transport)
        except Exception as e:                               # RCCL refuses duplicate devices on some stacks
            if rank == 0:
                np.savez(outfile, skip=np.array(repr(e)))
            return
        mine = lambda a: local_slab(a, cfg, rank, size)
        got = {}
        from ksfd_amd.dist import transport_selftest
        assert transport_selftest(ks, cfg, rank, size)            # the check open_handle runs on the RCCL transport
        got['rhs'] = gather_slabs(ks.rhs(mine(u)), cfg)
        got['jvp'] = gather_slabs(ks.jvp(mine(v), mine(u)), cfg)
        ks.set_state(mine(u))
        got['vmax'] = ks.velocity_max()
        got['worms'] = ks.count_worms()
        # start values generated on the device, slab by slab, from the global coarse samples
        zc = np.random.default_rng(5).normal(size=tuple(max(1, n // 4) for n in shape)) * 90.0
        ks.set_state_random(zc, 9000.0)
        got['start'] = gather_slabs(ks.get_state(), cfg)
        ks.set_state(mine(u))
        # assembled-Jacobian export: local rows, global (periodically wrapped) columns
        import scipy.sparse as sp
        rowptr, col, val = ks.jacobian_csr()
        A = sp.csr_matrix((val, col, rowptr), shape=(rowptr.size - 1, cfg.F * N))
        loc = (A @ v.reshape(cfg.F, N).T.reshape(-1)).reshape(-1, cfg.F).T.reshape(-1)
        got['csr_jv'] = gather_slabs(loc, cfg)
        opts = klib.default_step_opts(adapt=1, atol=0.01, rtol=1e-6, ksp_rtol=1e-11)
        t, h = 0.0, 0.02
        stats = []
        for _ in range(3):
            t, h, st, rc = ks.step(t, h, opts)
            stats.append((st.accepted, st.rejections, st.linear_its, st.wrms))
        # one very stiff fixed step: multigrid-preconditioned GMRES on the slabs (coarse levels exchange their own halos)
        stiff = klib.default_step_opts(adapt=0, atol=0.01, rtol=1e-6, ksp_rtol=1e-11, pc_type=1)
        t, hs_, st_mg, rc = ks.step(t, 5.0, stiff)
        got['mg_its'] = np.float64(st_mg.linear_its)
        got['state'] = gather_slabs(ks.get_state(), cfg)
        got['t'], got['h'] = t, h
        # ... and one more at a production tolerance: the V cycle then keeps its level vectors in fp32 (2-D; float halo rows travel
        # through the double-typed transport as half as many doubles)
        loose = klib.default_step_opts(adapt=0, atol=0.01, rtol=1e-6, ksp_rtol=1e-7, pc_type=1)
        t2, _, st_32, rc = ks.step(t, 5.0, loose)
        got['mg32_its'] = np.float64(st_32.linear_its)
        got['state32'] = gather_slabs(ks.get_state(), cfg)
        ks.close()
        if rank == 0:
            one = klib.KSFDHip(cfg)
            ref = {'rhs': one.rhs(u), 'jvp': one.jvp(v, u)}
            ref['csr_jv'] = ref['jvp']
            one.set_state_random(np.random.default_rng(5).normal(size=tuple(max(1, n // 4) for n in shape)) * 90.0, 9000.0)
            ref['start'] = one.get_state()
            one.set_state(u)
            ref['vmax'] = one.velocity_max()
            ref['worms'] = one.count_worms()
            t1, h1 = 0.0, 0.02
            for _ in range(3):
                t1, h1, st1, rc = one.step(t1, h1, opts)
            t1, hs1, st1mg, rc = one.step(t1, 5.0, stiff)
            ref['mg_its'] = np.float64(st1mg.linear_its)
            ref['state'], ref['t'], ref['h'] = one.get_state(), t1, h1
            t2, _, st1_32, rc = one.step(t1, 5.0, loose)
            ref['mg32_its'] = np.float64(st1_32.linear_its)
            ref['state32'] = one.get_state()
            one.close()
            np.savez(outfile, **{'got_' + k: np.asarray(got[k]) for k in got}, **{'ref_' + k: np.asarray(ref[k]) for k in ref})
    finally:
        dist.destroy_process_group()


def _run(size, shape, nlig, transport, tmp_path):
    outfile = str(tmp_path / 'result.npz')       # results come back through a file: a queue would block on join
    mp.spawn(_worker, args=(size, _free_port(), shape, nlig, transport, outfile), nprocs=size, join=True)
    z = np.load(outfile)
    if 'skip' in z:
        pytest.skip('transport %s unavailable here: %s' % (transport, z['skip']))
    assert np.array_equal(z['got_start'], z['ref_start'])
    for k in ('rhs', 'jvp', 'csr_jv'):
        assert rel_l2(z['got_' + k], z['ref_' + k]) < 1e-11, k
    assert rel_l2(z['got_state'], z['ref_state']) < 1e-9          # includes one h=5 step solved to ksp_rtol=1e-11
    assert rel_l2(z['got_state32'], z['ref_state32']) < 2e-6        # ... and one solved to 1e-7 through the fp32 V cycle (2-D)
    if len(shape) <= 2 or shape == (16, 16, 32):
        assert z['got_mg32_its'] <= 2 * z['ref_mg32_its'] + 8, (z['got_mg32_its'], z['ref_mg32_its'])
        # the slab hierarchy may be shallower than the single-rank one, never dramatically worse
        assert z['got_mg_its'] <= 2 * z['ref_mg_its'] + 8, (z['got_mg_its'], z['ref_mg_its'])
    for k in ('vmax', 'worms', 't', 'h'):
        assert np.allclose(z['got_' + k], z['ref_' + k], rtol=1e-9, atol=0), (k, z['got_' + k], z['ref_' + k])


@pytest.mark.parametrize('size,shape,nlig', [(2, (64, 48), 1), (3, (40, 36), 2), (2, (16, 12, 16), 1), (2, (33, 16), 1),
                                             (2, (140, 160), 1), (3, (64, 240), 2),    # >= 3 row segments per rank -> halo/compute overlap path
                                             (2, (16, 16, 32), 1),                     # 3-D with a 2-level multigrid hierarchy on the slabs
                                             (2, (96,), 2)])                           # 1-D slabs with the 1-D multigrid hierarchy
def test_slab_ranks_match_single_rank_host_transport(size, shape, nlig, tmp_path):
    _run(size, shape, nlig, 'host', tmp_path)


def test_slab_ranks_match_single_rank_rccl(tmp_path):
    _run(2, (64, 48), 1, 'rccl', tmp_path)


def _ring_of_one_case(transport, shape, nlig):
    """everything a slab rank does, on ONE rank that is its own ring neighbour, against the plain (wrap-index) handle"""
    from ksfd_amd import lib as klib
    from ksfd_amd.dist import open_self_ring, transport_selftest, reduction_selftest, spectral_selftest
    dim = len(shape)
    pow2 = all(n & (n - 1) == 0 for n in shape)
    cfg = ProblemConfig.standard(dim, shape, L=tuple(n * 4.0 / 1536 for n in shape), nlig=nlig)
    rng = np.random.default_rng(3)
    N = cfg.N
    rho = 9000 + 90 * rng.standard_normal(N)
    u = np.concatenate([rho] + [rho * cfg.lig_s[l] / cfg.lig_gamma[l] * (1 + 0.01 * rng.standard_normal(N)) for l in range(nlig)])
    v = rng.standard_normal(cfg.F * N)
    ks, keep = open_self_ring(cfg, 0, transport)
    one = klib.KSFDHip(cfg)
    try:
        # the checks open_handle runs on a new multi-rank handle
        assert transport_selftest(ks, cfg, 0, 1)
        assert reduction_selftest(ks, cfg, 0, 1)
        assert spectral_selftest(ks, cfg, 0, 1)
        assert rel_l2(ks.rhs(u), one.rhs(u)) < 1e-13
        assert rel_l2(ks.jvp(v, u), one.jvp(v, u)) < 1e-13
        ks.set_state(u), one.set_state(u)
        assert np.allclose(ks.velocity_max(), one.velocity_max(), rtol=1e-12, atol=0)
        assert abs(ks.count_worms() - one.count_worms()) <= 1e-12 * one.count_worms()
        if pow2 and dim >= 2:
            assert rel_l2(ks.spectral_apply(3.0, v), one.spectral_apply(3.0, v)) < 1e-5      # fp32 transforms, different summation order
        opts = klib.default_step_opts(adapt=1, atol=0.01, rtol=1e-6, ksp_rtol=1e-11)
        t, h, t1, h1 = 0.0, 0.02, 0.0, 0.02
        for _ in range(3):
            t, h, st, rc = ks.step(t, h, opts)
            t1, h1, st1, rc = one.step(t1, h1, opts)
            assert st.accepted == st1.accepted and st.rejections == st1.rejections
        assert np.allclose([t, h], [t1, h1], rtol=1e-9, atol=0)
        assert rel_l2(ks.get_state(), one.get_state()) < 1e-9
        if pow2 and dim >= 2:
            # a step through the slab-distributed spectral solver: all-to-all transposes whose every piece is this rank's own
            sp = klib.default_step_opts(adapt=0, atol=0.01, rtol=1e-6, ksp_rtol=1e-11, pc_type=4)
            t, h, st, rc = ks.step(t, 0.3, sp)
            t1, h1, st1, rc = one.step(t1, 0.3, sp)
            assert st.pc_used & 8 and st1.pc_used & 8
            assert rel_l2(ks.get_state(), one.get_state()) < 1e-9
        stiff = klib.default_step_opts(adapt=0, atol=0.01, rtol=1e-6, ksp_rtol=1e-11, pc_type=1)
        t, h, st, rc = ks.step(t, 5.0, stiff)
        t1, h1, st1, rc = one.step(t1, 5.0, stiff)
        assert rel_l2(ks.get_state(), one.get_state()) < 1e-9
        return ks.transport_name
    finally:
        ks.close()
        one.close()


RING1_CASES = [((64, 48), 1), ((64, 64), 2), ((140, 160), 1), ((32, 32, 32), 1), ((96,), 2)]


@pytest.mark.parametrize('shape,nlig', RING1_CASES)
def test_ring_of_one_rccl_transport_matches_wrap_index_handle(shape, nlig):
    """The RCCL transport EXECUTED on the one-GPU box: ksfd_dist{size = 1, transport = 1} -> ncclCommInitRank over one rank, ghost rows
    filled by grouped ncclSend/ncclRecv to itself (on the second stream, behind the interior row segments, where the grid has >= 3
    segments), ncclAllReduce + the k_publish hand-over of every reduction, the own-piece path of the spectral all-to-alls."""
    assert _ring_of_one_case('rccl', shape, nlig) == 'rccl-self'


@pytest.mark.parametrize('shape,nlig', RING1_CASES[:3])
def test_ring_of_one_host_transport_matches_wrap_index_handle(shape, nlig):
    assert _ring_of_one_case('host', shape, nlig) == 'host-self'


# The stage systems are solved to a tolerance; which iteration crosses it can differ between rank counts (different
# summation order), so a comparison tighter than the default ksp_rtol=1e-6 needs a tighter solve.
TIGHT = ('--petsc', '-ksp_rtol', '1e-11', '--')


def _solver_worker(rank, size, port, optfile, prefix):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(size), KSFD_DIST_BACKEND='gloo', KSFD_SHARE_GPU='1')
    from ksfd_amd import solver
    ts = solver.main('ksfd', optfile, '--save=' + prefix, *TIGHT)
    assert ts.getStepNumber() == 25 and not ts.diverged
    ts.cleanup()
    dist.destroy_process_group()


def test_solver_main_on_two_ranks_matches_one_rank(tmp_path):
    """`torchrun -m ksfd_amd.solver @options` (here: 2 ranks sharing cuda:0 over gloo): per-rank series files
    <prefix>s2r<rank>.* whose slabs, put side by side, equal the single-rank run."""
    from conftest import GOLDEN
    from ksfd_amd import solver
    from ksfd_amd.timeseries import read_series
    optfile = '@' + os.path.join(GOLDEN, 'options', 'ks2d_two_ligands.txt')
    p2 = str(tmp_path / 'two' / 'run')
    mp.spawn(_solver_worker, args=(2, _free_port(), optfile, p2), nprocs=2, join=True)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE'):
        os.environ.pop(k, None)
    p1 = str(tmp_path / 'one' / 'run')
    ts = solver.main('ksfd', optfile, '--save=' + p1, *TIGHT)
    ts.cleanup()
    one = read_series(p1)
    parts = [read_series(p2, size=2, rank=r) for r in range(2)]
    dt_rel = np.abs(parts[0]['times'] - one['times']).max() / one['times'].max()
    assert dt_rel < 1e-8, dt_rel
    both = np.concatenate([parts[0]['data'][25], parts[1]['data'][25]], axis=2)      # (dof, nx, ny): slabs along y
    assert both.shape == one['data'][25].shape
    assert rel_l2(both, one['data'][25]) < 1e-8


def _spectral_worker(rank, size, port, shape, nlig, outfile):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=size)
    try:
        from ksfd_amd import lib as klib
        from ksfd_amd.dist import open_handle, local_slab, gather_slabs
        L = tuple(n * 4.0 / 1536 for n in shape)
        if len(shape) == 3:
            cfg = ProblemConfig.standard(3, shape, L=L, nlig=nlig)
        elif nlig == 3:          # two ligands sharing a group + a repellent: F = 4, two complex pairs
            cfg = ProblemConfig(dim=2, n=shape, L=L, lig_group=[0, 0, 1], lig_w=[1.0, 0.5, 1.0], lig_s=[0.01, 0.02, 0.001],
                                lig_gamma=[0.01, 0.03, 0.001], lig_D=[1e-6, 3e-6, 1e-5], grp_alpha=[1500.0, 1500.0], grp_beta=[5.56e-4, -5.56e-4])
        else:
            cfg = ProblemConfig.standard(2, shape, L=L, nlig=nlig)
        rng = np.random.default_rng(3)
        N = cfg.N
        rho = 9000 + 90 * rng.standard_normal(N)
        u = np.concatenate([rho] + [rho * cfg.lig_s[l] / cfg.lig_gamma[l] * (1 + 0.01 * rng.standard_normal(N)) for l in range(nlig)])
        v = rng.standard_normal(cfg.F * N)
        ks, keep = open_handle(cfg, rank, size, 0, transport='host')
        mine = lambda a: local_slab(a, cfg, rank, size)
        got = {}
        from ksfd_amd.dist import spectral_selftest
        assert spectral_selftest(ks, cfg, rank, size)             # the check open_handle runs on the RCCL transport
        ks.set_state(mine(u))
        got['spec'] = gather_slabs(ks.spectral_apply(3.0, mine(v)), cfg)
        opts = klib.default_step_opts(adapt=0, atol=0.01, rtol=1e-6, ksp_rtol=1e-11, pc_type=4)
        t, h, st, rc = ks.step(0.0, 0.3, opts)
        got['pc'] = np.float64(st.pc_used)
        got['its'] = np.float64(st.linear_its)
        got['state'] = gather_slabs(ks.get_state(), cfg)
        ks.close()
        if rank == 0:
            one = klib.KSFDHip(cfg)
            one.set_state(u)
            ref = {'spec': one.spectral_apply(3.0, v)}
            t, h, st1, rc = one.step(0.0, 0.3, opts)
            ref['state'], ref['its'] = one.get_state(), np.float64(st1.linear_its)
            one.close()
            np.savez(outfile, **{'got_' + k: np.asarray(got[k]) for k in got}, **{'ref_' + k: np.asarray(ref[k]) for k in ref})
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('size,shape,nlig,split', [(2, (64, 64), 1, False), (4, (128, 64), 2, False), (2, (32, 256), 3, False), (2, (64, 128), 2, True),
                                                   (4, (32, 64), 3, True)])
def test_slab_distributed_spectral_solver_matches_single_rank(size, shape, nlig, split, tmp_path, monkeypatch):
    """the y-transforms of the spectral preconditioner across slab ranks (all-to-all transposes, columns of {kx, -kx} pairs kept on
    one rank): same operator as on one rank (fp32 FFTs: summation order differs, 1e-5), and a step solved with it is the single-rank
    step to the solver tolerance"""
    if split:       # the two-phase column kernel (field pairs that do not fit the LDS together), forced on a small grid; the ranks inherit the knob
        monkeypatch.setenv('KSFD_SPEC_SPLIT', '1')
    outfile = str(tmp_path / 'result.npz')
    mp.spawn(_spectral_worker, args=(size, _free_port(), shape, nlig, outfile), nprocs=size, join=True)
    z = np.load(outfile)
    assert int(z['got_pc']) & 8                                   # the spectral solver really ran on the slabs
    assert rel_l2(z['got_spec'], z['ref_spec']) < 1e-5
    assert rel_l2(z['got_state'], z['ref_state']) < 1e-9
    assert z['got_its'] <= z['ref_its'] + 4


@pytest.mark.parametrize('size,shape,nlig', [(2, (32, 32, 32), 1), (4, (64, 32, 64), 2), (2, (32, 64, 32), 1)])
def test_slab_distributed_spectral_solver_3d_matches_single_rank(size, shape, nlig, tmp_path):
    """3-D on z slabs (BASELINE configs[4] is 512^3 on 8 GPUs): x and y transforms local, the z transforms after an all-to-all over the
    x positions (columns (kx,ky) and (-kx,-ky) on one rank, a column = P pieces of nz/P points): same operator as on one rank, and
    a step solved with it is the single-rank step to the solver tolerance"""
    outfile = str(tmp_path / 'result.npz')
    mp.spawn(_spectral_worker, args=(size, _free_port(), shape, nlig, outfile), nprocs=size, join=True)
    z = np.load(outfile)
    assert int(z['got_pc']) & 8
    assert rel_l2(z['got_spec'], z['ref_spec']) < 1e-5
    assert rel_l2(z['got_state'], z['ref_state']) < 1e-9
    assert z['got_its'] <= z['ref_its'] + 4
