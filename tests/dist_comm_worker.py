"""Worker of tests/test_comm_gpu.py: one rank of a run whose plate sums go through the
library's own RCCL communicator (vmp_comm_init_rank / vmp_allreduce_sum_f64).  Launched by
``python -m torch.distributed.run`` exactly like the driver launches bench.py; with one GPU the
world has one rank (RCCL refuses two ranks on one device), with more it has one rank per GPU."""
import ctypes
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def main():
    golden, out = sys.argv[1], sys.argv[2]
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    local = int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local % torch.cuda.device_count())
    dist.init_process_group('nccl', device_id=torch.device('cuda', local % torch.cuda.device_count()))
    import bayespy_amd.nodes as nodes
    from bayespy_amd.inference import VB
    from bayespy_amd.device import get_runtime, ptr
    from models import build_pca
    rt = get_runtime()
    res = {}
    # ---- the raw entry points ------------------------------------------------------------------
    t = torch.arange(1000, dtype=torch.float64, device=rt.device) * (rank + 1)
    rt.all_reduce_sum_(t)
    assert rt._comm_state is True, 'the library communicator was not created'
    r, w = ctypes.c_int32(-1), ctypes.c_int32(-1)
    rt.check(rt.lib.vmp_comm_info(rt.ctx, ctypes.byref(r), ctypes.byref(w)))
    assert (r.value, w.value) == (rank, world)
    res['allreduce'] = t.cpu().numpy()
    res['expect'] = np.arange(1000.0) * (world * (world + 1) // 2)
    # a strided (non-contiguous) view and ordering against kernels on the same stream
    m = torch.ones(64, 8, dtype=torch.float64, device=rt.device)
    v = m[:, 3]
    v.mul_(rank + 2.0)
    rt.all_reduce_sum_(v)
    res['strided'] = m.cpu().numpy()
    # ---- a fused PCA block with the plate split over the ranks, both statistics forms --------------
    g = np.load(os.path.join(golden, 'pca_n777_d20_k5.npz'))
    y, x0 = g['y'], g['x0']
    N = y.shape[1]
    lo, hi = N * rank // world, N * (rank + 1) // world
    for stats in ('gram', 'stream'):
        Q = build_pca(nodes, VB, y[:, lo:hi], x0[lo:hi], x0.shape[1], shard=True)
        Q.plans[0].stats = stats
        Q.update(repeat=int(g['n_iter']), verbose=False)
        res['L_' + stats] = np.array(Q.L[:Q.iter])
        res['W_' + stats] = np.asarray(Q['W'].u[0])
    # ---- the generic engine on the same split: its sweep is recorded into a HIP graph WITH the
    # library's all-reduces inside (on a one-GPU box the world has one rank: the sharded code
    # path, every collective included, is switched on for it) ------------------------------------
    if world == 1:
        os.environ['BAYESPY_AMD_SHARD_WORLD1'] = '1'
    n_it = max(int(g['n_iter']), 8)
    for mode in ('graph', 'eager'):
        os.environ['BAYESPY_AMD_GRAPH'] = '1' if mode == 'graph' else '0'
        before = rt.collective_calls['library']
        Q = build_pca(nodes, VB, y[:, lo:hi], x0[lo:hi], x0.shape[1], shard=True, engine='generic')
        Q.ignore_bound_checks = True
        Q.update(repeat=n_it, verbose=False)
        res['Lg_' + mode] = np.array(Q.L[:Q.iter])
        res['Wg_' + mode] = np.asarray(Q['W'].u[0])
        info = Q.plans[0].graph_info()
        res['recorded_' + mode] = np.array([int(info['recorded']), int(info['replays'])])
        res['calls_' + mode] = np.array([rt.collective_calls['library'] - before])
    os.environ.pop('BAYESPY_AMD_SHARD_WORLD1', None)
    os.environ['BAYESPY_AMD_GRAPH'] = '1'
    res['lo'], res['hi'] = lo, hi
    np.savez(os.path.join(out, 'rank%d.npz' % rank), **res)
    rt.check(rt.lib.vmp_comm_destroy(rt.ctx))
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
