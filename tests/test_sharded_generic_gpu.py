"""
GPU: the generic engine with a SHARDED observation plate (Node.shard): two ranks, each
holding half of the plate, must reproduce the single-process reference traces -- messages
to the replicated nodes and the lower-bound terms are completed by all-reduce.

The ranks are launched with ``torch.distributed.run`` exactly like the driver launches
bench.py.  On a one-GPU box both ranks share the device and the collective backend is gloo
(RCCL refuses two ranks on one device); with two or more GPUs the ranks take one GPU each and
the collective is the library's RCCL all-reduce (backend nccl, chosen automatically).
"""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def _launch(case, golden_dir, tmp_path, port):
    env = dict(os.environ)
    import torch
    # two GPUs or more: one rank per GPU over RCCL; one GPU: both ranks share it over gloo
    env.setdefault('VMP_TEST_BACKEND', 'nccl' if torch.cuda.device_count() >= 2 else 'gloo')
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
           '--master-addr', '127.0.0.1', '--master-port', str(port),
           os.path.join(HERE, 'dist_generic_worker.py'), case, golden_dir, str(tmp_path)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return [np.load(os.path.join(str(tmp_path), 'rank%d.npz' % k)) for k in range(2)]


def test_sharded_masked_pca_matches_reference(golden_dir, tmp_path):
    r0, r1 = _launch('masked_pca', golden_dir, tmp_path, 29541)
    g = np.load(os.path.join(golden_dir, 'small_models.npz'))
    for r in (r0, r1):
        np.testing.assert_allclose(r['L'], g['mpca_L'], rtol=1e-9)
        np.testing.assert_allclose(r['W_u0'], g['mpca_W_u0'], rtol=1e-7, atol=1e-10)
        np.testing.assert_allclose(r['tau_u0'], g['mpca_tau_u0'], rtol=1e-8)
        np.testing.assert_allclose(r['alpha_u0'], g['mpca_alpha_u0'], rtol=1e-8)
        np.testing.assert_allclose(r['X_u0'], g['mpca_X_u0'][:, int(r['lo']):int(r['hi'])],
                                   rtol=1e-7, atol=1e-10)
    # replicated nodes are bitwise identical on the two ranks
    assert np.array_equal(r0['W_u0'], r1['W_u0']) and np.array_equal(r0['L'], r1['L'])


def test_sharded_fused_masked_pca_matches_reference(golden_dir, tmp_path):
    r0, r1 = _launch('masked_pca_fused', golden_dir, tmp_path, 29548)
    g = np.load(os.path.join(golden_dir, 'masked_pca.npz'))
    for r in (r0, r1):
        np.testing.assert_allclose(r['L'], g['m2_L'], rtol=1e-9)
        np.testing.assert_allclose(r['W_u0'], g['m2_W_u0'], rtol=1e-7, atol=1e-10)
        np.testing.assert_allclose(r['tau_u'], g['m2_tau_u'], rtol=1e-8)
        np.testing.assert_allclose(r['X_u0'], g['m2_X_u0'][:, int(r['lo']):int(r['hi'])],
                                   rtol=1e-7, atol=1e-10)
    assert np.array_equal(r0['W_u0'], r1['W_u0']) and np.array_equal(r0['L'], r1['L'])


@pytest.mark.parametrize('engine,port', [('fused', 29550), ('generic', 29551)])
def test_sharded_erasure_patterns_match_reference(golden_dir, tmp_path, engine, port):
    """Dimensions without observations under a sharded plate: "ignored" is a global property (the
    observation counts per dimension are summed over the ranks)."""
    r0, r1 = _launch('erasures_' + engine, golden_dir, tmp_path, port)
    g = np.load(os.path.join(golden_dir, 'masked_pca_erasures.npz'))
    for r in (r0, r1):
        assert str(r['engine']) == ('MaskedPCAPlan' if engine == 'fused' else 'GenericPlan')
        np.testing.assert_allclose(r['L'], g['e1_L'], rtol=1e-9)
        np.testing.assert_allclose(r['W_u0'], g['e1_W_u0'], rtol=1e-7, atol=1e-10)
        np.testing.assert_allclose(r['alpha_u0'], g['e1_alpha_u0'], rtol=1e-8)
        np.testing.assert_allclose(r['X_u0'], g['e1_X_u0'][:, int(r['lo']):int(r['hi'])],
                                   rtol=1e-7, atol=1e-10)
    assert np.array_equal(r0['W_u0'], r1['W_u0']) and np.array_equal(r0['L'], r1['L'])


def test_a_rank_without_any_plate_element(golden_dir, tmp_path):
    """Fewer plate elements than ranks: rank 0 of the two holds an EMPTY local plate in each of the
    four fused blocks; both ranks reproduce the single-process oracle run on all the data."""
    from oracle.pca import PCAOracle
    from oracle.masked_pca import MaskedPCAOracle
    from oracle.gmm import GMMOracle
    from oracle.lssm import LSSMOracle
    r0, r1 = _launch('empty_rank', golden_dir, tmp_path, 29552)
    g = r1
    oracles = dict(
        pca=PCAOracle(g['in_y'], g['in_x0']),
        mpca=MaskedPCAOracle(np.where(g['in_mask'], g['in_y'], np.nan), g['in_mask'], g['in_x0']),
        gmm=GMMOracle(g['in_yg'], g['in_lab'], 2),
        lssm=LSSMOracle(g['in_yl'], g['in_xl'], g['in_cl']))
    engines = dict(pca='PCAPlan', mpca='MaskedPCAPlan', gmm='GMMPlan', lssm='LSSMPlan')
    for nm, o in oracles.items():
        o.iterate(3)
        for r in (r0, r1):
            assert str(r[nm + '_engine']) == engines[nm]
            np.testing.assert_allclose(r[nm + '_L'], np.array(o.L), rtol=1e-9, err_msg=nm)
        assert np.array_equal(r0[nm + '_L'], r1[nm + '_L'])


def test_sharded_rotation_matches_reference(golden_dir, tmp_path):
    r0, r1 = _launch('rotation', golden_dir, tmp_path, 29542)
    g = np.load(os.path.join(golden_dir, 'rotations.npz'))
    for r in (r0, r1):
        np.testing.assert_allclose(r['L_before'], g['rotm_L_before'], rtol=1e-9)
        np.testing.assert_allclose(r['L_after'], g['rotm_L_after'], rtol=1e-7)
        np.testing.assert_allclose(r['W_u0_rot'], g['rotm_W_u0_rot'], rtol=1e-6, atol=1e-8)
        np.testing.assert_allclose(r['X_u0_rot'],
                                   g['rotm_X_u0_rot'][:, int(r['lo']):int(r['hi'])],
                                   rtol=1e-6, atol=1e-8)


def test_sharded_lssm_matches_reference(golden_dir, tmp_path):
    """Batched linear state-space model with the sequence plate split over two ranks
    (BASELINE config 5 shards the sequences); only the chain X is declared sharded."""
    r0, r1 = _launch('lssm', golden_dir, tmp_path, 29543)
    g = np.load(os.path.join(golden_dir, 'lssm.npz'))
    for r in (r0, r1):
        assert str(r['engine']) == 'LSSMPlan'          # the fused state-space block's real kernels
        np.testing.assert_allclose(r['L'], g['lssmB_L'], rtol=1e-9)
        for nm in ('A', 'C', 'tau', 'alpha', 'gamma', 'nu'):
            np.testing.assert_allclose(r['L_' + nm], g['lssmB_%s_L' % nm], rtol=1e-8, atol=1e-7,
                                       err_msg=nm)
    assert np.array_equal(r0['A_u0'], r1['A_u0'])


@pytest.mark.parametrize('case,port', [('pca_fused_gram', 29553), ('pca_fused_stream', 29554),
                                       ('pca_generic', 29555)])
def test_two_ranks_through_the_pca_block(golden_dir, tmp_path, case, port):
    """Two real ranks through the headline block's REAL kernels (both statistics forms) and
    through the generic engine's fused node update, the plate split a third / two thirds: bounds
    and moments of the unsharded live-reference trace, replicated nodes bitwise equal."""
    r0, r1 = _launch(case, golden_dir, tmp_path, port)
    g = np.load(os.path.join(golden_dir, 'pca_n777_d20_k5.npz'))
    for r in (r0, r1):
        assert str(r['engine']) == ('GenericPlan' if case == 'pca_generic' else 'PCAPlan')
        assert int(r['calls'][0]) > 0
        np.testing.assert_allclose(r['L'], g['L'], rtol=1e-9)
        for nm in ('Y', 'X', 'W', 'tau', 'alpha'):
            np.testing.assert_allclose(r['L_' + nm], g['L_' + nm], rtol=1e-8, atol=1e-7, err_msg=nm)
        np.testing.assert_allclose(r['W_u0'], g['W_u0'], rtol=1e-7, atol=1e-10)
        np.testing.assert_allclose(r['W_u1'], g['W_u1'], rtol=1e-7, atol=1e-10)
        np.testing.assert_allclose(r['tau_u0'], g['tau_u0'], rtol=1e-9)
        np.testing.assert_allclose(r['alpha_u1'], g['alpha_u1'], rtol=1e-8)
        np.testing.assert_allclose(r['X_u0'], g['X_u0'][:, int(r['lo']):int(r['hi'])],
                                   rtol=1e-7, atol=1e-10)
    assert int(r0['hi']) - int(r0['lo']) != int(r1['hi']) - int(r1['lo'])
    for k in ('L', 'W_u0', 'W_u1', 'tau_u0', 'alpha_u1'):
        assert np.array_equal(r0[k], r1[k]), k


@pytest.mark.parametrize('case,port', [('gmm_fused', 29556), ('gmm_generic', 29557)])
def test_two_ranks_through_the_mixture_block(golden_dir, tmp_path, case, port):
    r0, r1 = _launch(case, golden_dir, tmp_path, port)
    g = np.load(os.path.join(golden_dir, 'gmm_n3000_d8_k16.npz'))
    for r in (r0, r1):
        assert str(r['engine']) == ('GenericPlan' if case == 'gmm_generic' else 'GMMPlan')
        assert int(r['calls'][0]) > 0
        np.testing.assert_allclose(r['L'], g['L'], rtol=1e-9)
        for nm in ('Y', 'mu', 'Lambda', 'z', 'alpha'):
            np.testing.assert_allclose(r['L_' + nm], g['L_' + nm], rtol=1e-8, atol=1e-6, err_msg=nm)
        np.testing.assert_allclose(r['mu_u0'], g['mu_u0'], rtol=1e-7, atol=1e-10)
        np.testing.assert_allclose(r['Lambda_u0'], g['Lambda_u0'], rtol=1e-7, atol=1e-10)
        np.testing.assert_allclose(r['alpha_u0'], g['alpha_u0'], rtol=1e-7)
        np.testing.assert_allclose(r['z_u0'], g['z_u0'][int(r['lo']):int(r['hi'])], rtol=1e-6,
                                   atol=1e-12)
    for k in ('L', 'mu_u0', 'Lambda_u0', 'alpha_u0'):
        assert np.array_equal(r0[k], r1[k]), k


def test_sharded_lssm_rotation_matches_reference(golden_dir, tmp_path):
    """The state-space rotation on the fused block with the sequences split over two ranks: same
    bound before / after and the same rotated moments as the single-process reference run."""
    r0, r1 = _launch('lssm_rotation', golden_dir, tmp_path, 29549)
    g = np.load(os.path.join(golden_dir, 'lssm_rotations.npz'))
    for r in (r0, r1):
        assert str(r['engine']) == 'LSSMPlan'
        np.testing.assert_allclose(r['L_before'], g['batch_L_before'], rtol=1e-9)
        np.testing.assert_allclose(r['L_after'], g['batch_L_after'], rtol=1e-6)
        np.testing.assert_allclose(r['A_u0_rot'], g['batch_A_u0_rot'], rtol=1e-5, atol=1e-6)
        lo, hi = int(r['lo']), int(r['hi'])
        ref = g['batch_X_u0_rot'][lo:hi]
        np.testing.assert_allclose(r['X_u0_rot'], ref, rtol=1e-5, atol=1e-6 * np.abs(ref).max())
    assert np.array_equal(r0['A_u0_rot'], r1['A_u0_rot'])


def test_sharded_hidden_markov_chains_match_reference(golden_dir, tmp_path):
    """A batch of hidden Markov chains with the chain plate split over two ranks: only the
    initial-state node is declared sharded, the chain, its categorical view and the mixture
    inherit the partition; messages to the replicated transition and emission parameters and
    the bound terms of the sharded nodes are all-reduced."""
    r0, r1 = _launch('hmm', golden_dir, tmp_path, 29547)
    g = np.load(os.path.join(golden_dir, 'markov_chains.npz'))
    for r in (r0, r1):
        np.testing.assert_allclose(r['L'], g['hmm3_L'], rtol=1e-9)
        np.testing.assert_allclose(r['m_u0'], g['hmm3_m_u_0'], rtol=1e-7)
        np.testing.assert_allclose(r['t_u0'], g['hmm3_t_u_0'], rtol=1e-7)
        np.testing.assert_allclose(r['A_u0'], g['hmm3_A_u_0'], rtol=1e-7)
        lo, hi = int(r['lo']), int(r['hi'])
        np.testing.assert_allclose(r['Z_u0'], g['hmm3_Z_u_0'][lo:hi], rtol=1e-7, atol=1e-12)
    assert np.array_equal(r0['m_u0'], r1['m_u0'])


def test_sharded_masked_lssm_matches_reference(golden_dir, tmp_path):
    """The state-space block with array masks on the device kernels, the sequence plate split
    2 + 3 over two ranks: the unsharded live-reference trace on both ranks, each rank's own <x>."""
    r0, r1 = _launch('lssm_masked', golden_dir, tmp_path, 29551)
    g = np.load(os.path.join(golden_dir, 'lssm_masked.npz'))
    for r in (r0, r1):
        assert str(r['engine']) == 'MaskedLSSMPlan'
        np.testing.assert_allclose(r['L'], g['mb_L'], rtol=1e-9)
        np.testing.assert_allclose(r['C_u0'], g['mb_C_u0'], rtol=1e-7, atol=1e-9)
        np.testing.assert_allclose(r['A_u0'], g['mb_A_u0'], rtol=1e-7, atol=1e-9)
        lo, hi = int(r['lo']), int(r['hi'])
        np.testing.assert_allclose(r['X_u0'], g['mb_X_u0'][lo:hi], rtol=1e-7, atol=1e-9)
    assert np.array_equal(r0['C_u0'], r1['C_u0'])
