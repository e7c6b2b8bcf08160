"""
GPU: the library's own RCCL communicator (vmp_comm_* / vmp_allreduce_sum_f64).

The ranks are launched with ``torch.distributed.run`` and the real ``nccl`` (= RCCL) backend,
exactly like the driver launches bench.py: one rank per visible GPU (a world of one rank on a
one-GPU box -- RCCL initialisation, the id hand-over, stream ordering against the library's
kernels and the ``device_id`` path are exercised all the same).
"""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def test_rccl_communicator_in_context(golden_dir, tmp_path):
    import torch
    world = max(1, min(torch.cuda.device_count(), 8))
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node',
           str(world), '--master-addr', '127.0.0.1', '--master-port', '29561',
           os.path.join(HERE, 'dist_comm_worker.py'), golden_dir, str(tmp_path)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    g = np.load(os.path.join(golden_dir, 'pca_n777_d20_k5.npz'))
    outs = [np.load(os.path.join(str(tmp_path), 'rank%d.npz' % k)) for k in range(world)]
    for k, o in enumerate(outs):
        np.testing.assert_array_equal(o['allreduce'], o['expect'])
        col = float(sum(range(2, world + 2)))
        expect = np.ones((64, 8))
        expect[:, 3] = col
        np.testing.assert_array_equal(o['strided'], expect)
        for stats in ('gram', 'stream'):
            np.testing.assert_allclose(o['L_' + stats], g['L'], rtol=1e-9)
            np.testing.assert_allclose(o['W_' + stats], g['W_u0'], rtol=1e-7, atol=1e-10)
            # replicated nodes are bitwise identical on all ranks
            np.testing.assert_array_equal(o['W_' + stats], outs[0]['W_' + stats])
        # the generic engine's sharded sweep: recorded with the RCCL all-reduces inside, replayed,
        # same bounds as the eager sweeps and as the unsharded live reference
        n = len(g['L'])
        assert o['recorded_graph'][0] == 1 and o['recorded_graph'][1] >= 3, o['recorded_graph']
        assert o['recorded_eager'][0] == 0
        assert o['calls_graph'][0] > 0 and o['calls_eager'][0] > o['calls_graph'][0]
        np.testing.assert_allclose(o['Lg_graph'][:n], g['L'], rtol=1e-9)
        np.testing.assert_allclose(o['Lg_graph'], o['Lg_eager'], rtol=1e-12)
        np.testing.assert_allclose(o['Wg_graph'], o['Wg_eager'], rtol=1e-10, atol=1e-12)
        np.testing.assert_array_equal(o['Wg_graph'], outs[0]['Wg_graph'])
