import faulthandler, os, sys, socket, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp

def worker(rank, size, port, transport):
    logf = open('gpurun_out/dbg_dist_rank%d.log' % rank, 'w', buffering=1)
    faulthandler.dump_traceback_later(40, repeat=False, file=logf)
    def log(*a):
        print(time.strftime('%H:%M:%S'), *a, file=logf, flush=True)
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=size)
    log('pg up')
    from ksfd_amd import lib as klib
    from ksfd_amd.config import ProblemConfig
    from ksfd_amd.dist import open_handle, local_slab, gather_slabs
    cfg = ProblemConfig.standard(2, (64, 96), L=(0.2, 0.25), nlig=1)
    rng = np.random.default_rng(3); N = cfg.N
    rho = 9000 + 90 * rng.standard_normal(N)
    u = np.concatenate([rho, rho + rng.standard_normal(N)])
    ks, keep = open_handle(cfg, rank, size, 0, transport=transport)
    log('handle open', ks.slow_range)
    r = ks.rhs(local_slab(u, cfg, rank, size)); log('rhs done', float(np.abs(r).max()))
    g = gather_slabs(r, cfg); log('gather done')
    ks.set_state(local_slab(u, cfg, rank, size)); log('state set')
    log('vmax', ks.velocity_max())
    log('worms', ks.count_worms())
    opts = klib.default_step_opts(adapt=1, atol=0.01, rtol=1e-6, ksp_rtol=1e-11)
    t, h = 0.0, 0.02
    for i in range(3):
        t, h, st, rc = ks.step(t, h, opts, raise_on_error=False)
        log('step', i, t, h, st.accepted, st.rejections, st.linear_its, st.wrms, rc, ks.last_error() if rc else '')
    stiff = klib.default_step_opts(adapt=0, atol=0.01, rtol=1e-6, ksp_rtol=1e-11, pc_type=1)
    t, hh, st, rc = ks.step(t, 5.0, stiff, raise_on_error=False)
    log('mg step', t, st.accepted, st.linear_its, st.wrms, rc, ks.last_error() if rc else '')
    ks.close(); log('closed')
    dist.destroy_process_group(); log('done')

if __name__ == '__main__':
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(worker, args=(2, port, sys.argv[1] if len(sys.argv) > 1 else 'host'), nprocs=2, join=True)
    print('spawn joined')
