"""Worker of tests/test_sharded_generic_gpu.py: one rank of a sharded run of the generic
engine.  Launched by ``python -m torch.distributed.run``; all ranks may share one GPU
(backend from VMP_TEST_BACKEND, gloo on a one-GPU box; nccl = RCCL on a multi-GPU node)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def main():
    case, golden, out = sys.argv[1], sys.argv[2], sys.argv[3]
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    ndev = torch.cuda.device_count()
    torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')) % ndev)
    backend = os.environ.get('VMP_TEST_BACKEND', 'gloo')
    if backend == 'nccl':
        dist.init_process_group('nccl', device_id=torch.device('cuda', torch.cuda.current_device()))
    else:
        dist.init_process_group(backend)
    import bayespy_amd.nodes as nodes
    from bayespy_amd.inference import VB, transformations
    res = {}
    if case == 'masked_pca':
        g = np.load(os.path.join(golden, 'small_models.npz'))
        y, mask, x0 = g['mpca_y'], g['mpca_mask'], g['mpca_x0']
        D, N = y.shape
        K = x0.shape[1]
        lo, hi = N * rank // world, N * (rank + 1) // world
        alpha = nodes.Gamma(1e-2, 1e-2, plates=(K,), name='alpha')
        W = nodes.GaussianARD(0, alpha, shape=(K,), plates=(D, 1), name='W')
        X = nodes.GaussianARD(0, 1, shape=(K,), plates=(1, hi - lo), name='X').shard(-1)
        F = nodes.SumMultiply('i,i', W, X, name='F').shard(-1)
        tau = nodes.Gamma(1e-2, 1e-2, name='tau')
        Y = nodes.GaussianARD(F, tau, name='Y').shard(-1)
        X.initialize_from_value(x0[None, lo:hi])
        Y.observe(y[:, lo:hi], mask=mask[:, lo:hi])
        Q = VB(Y, F, W, X, tau, alpha, engine='generic')
        Q.ignore_bound_checks = True
        Q.update(repeat=4, verbose=False)
        res['L'] = np.array(Q.L[:Q.iter])
        res['W_u0'], res['tau_u0'] = np.asarray(W.u[0]), np.asarray(tau.u[0])
        res['alpha_u0'] = np.asarray(alpha.u[0])
        res['X_u0'] = np.asarray(X.u[0])
        res['lo'], res['hi'] = lo, hi
    elif case == 'masked_pca_fused':
        # the fused missing-data block with the plate split over the ranks: M_d, r_d and the
        # scalar sums are all-reduced after every X pass
        from models import build_masked_pca
        g = np.load(os.path.join(golden, 'masked_pca.npz'))
        y, mask, x0 = g['in_m2_y'], g['in_m2_mask'], g['in_m2_x0']
        N = y.shape[1]
        lo, hi = N * rank // world, N * (rank + 1) // world
        Q = build_masked_pca(nodes, VB, y[:, lo:hi], mask[:, lo:hi], x0[lo:hi], shard=True)
        assert type(Q.plans[0]).__name__ == 'MaskedPCAPlan'
        Q.update(repeat=len(g['m2_L']), verbose=False)
        res['L'] = np.array(Q.L[:Q.iter])
        res['W_u0'] = np.asarray(Q['W'].u[0])
        res['X_u0'] = np.asarray(Q['X'].u[0])
        res['tau_u'] = np.array([np.asarray(u) for u in Q['tau'].u], dtype=np.float64)
        res['lo'], res['hi'] = lo, hi
    elif case in ('erasures_fused', 'erasures_generic'):
        # a dimension observed on ONE rank only is not an ignored plate of W on the other, a dimension
        # observed nowhere is one on both (tests/golden/masked_pca_erasures.npz, case e1)
        from models import build_masked_pca
        g = np.load(os.path.join(golden, 'masked_pca_erasures.npz'))
        y, mask, x0 = g['in_e1_y'], g['in_e1_mask'], g['in_e1_x0']
        N = y.shape[1]
        lo, hi = N * rank // world, N * (rank + 1) // world
        kw = {'engine': 'generic'} if case == 'erasures_generic' else {}
        Q = build_masked_pca(nodes, VB, y[:, lo:hi], mask[:, lo:hi], x0[lo:hi], shard=True, **kw)
        res['engine'] = type(Q.plans[0]).__name__
        Q.update(repeat=len(g['e1_L']), verbose=False)
        res['L'] = np.array(Q.L[:Q.iter])
        res['W_u0'] = np.asarray(Q['W'].u[0])
        res['alpha_u0'] = np.asarray(Q['alpha'].u[0])
        res['X_u0'] = np.asarray(Q['X'].u[0])
        res['lo'], res['hi'] = lo, hi
    elif case == 'empty_rank':
        # fewer plate elements than ranks: rank 0 holds NOTHING, rank 1 everything; every fused
        # block must take part in the collectives with zero statistics
        from models import build_pca, build_masked_pca
        from test_gmm_gpu import _build as build_gmm
        from test_lssm_gpu import _build as build_lssm
        rs = np.random.RandomState(11)
        D, K = 5, 2
        n = 0 if rank == 0 else 3
        y, x0 = rs.normal(size=(D, 3)), rs.normal(size=(3, K))
        Q = build_pca(nodes, VB, y[:, :n], x0[:n], K, shard=True)
        Q.update(repeat=3, verbose=False)
        res['pca_L'], res['pca_engine'] = np.array(Q.L[:3]), type(Q.plans[0]).__name__
        mask = rs.rand(D, 3) < 0.8
        Q = build_masked_pca(nodes, VB, y[:, :n], mask[:, :n], x0[:n], shard=True)
        Q.update(repeat=3, verbose=False)
        res['mpca_L'], res['mpca_engine'] = np.array(Q.L[:3]), type(Q.plans[0]).__name__
        yg, lab = rs.normal(size=(3, 2)), np.array([0, 1, 1])
        Q = build_gmm(yg[:n], lab[:n], 2, shard=True)
        Q.update(repeat=3, verbose=False)
        res['gmm_L'], res['gmm_engine'] = np.array(Q.L[:3]), type(Q.plans[0]).__name__
        yl, xl, cl = rs.normal(size=(2, 3, 6)), rs.normal(size=(3, 6, 2)), rs.normal(size=(2, 2))
        Q = build_lssm(yl[:, :n], xl[:n], cl, False, shard=True)
        Q.update(repeat=3, verbose=False)
        res['lssm_L'], res['lssm_engine'] = np.array(Q.L[:3]), type(Q.plans[0]).__name__
        for k_, v_ in dict(y=y, x0=x0, mask=mask, yg=yg, lab=lab, yl=yl, xl=xl, cl=cl).items():
            res['in_' + k_] = v_
    elif case == 'rotation':
        g = np.load(os.path.join(golden, 'rotations.npz'))
        y, mask, x0 = g['rotm_y'], g['rotm_mask'], g['rotm_x0']
        D, N = y.shape
        K = x0.shape[1]
        lo, hi = N * rank // world, N * (rank + 1) // world
        alpha = nodes.Gamma(1e-2, 1e-2, plates=(K,), name='alpha')
        W = nodes.GaussianARD(0, alpha, shape=(K,), plates=(D, 1), name='W')
        X = nodes.GaussianARD(0, 1, shape=(K,), plates=(1, hi - lo), name='X').shard(-1)
        F = nodes.SumMultiply('i,i', W, X, name='F')      # inherits the partition
        tau = nodes.Gamma(1e-2, 1e-2, name='tau')
        Y = nodes.GaussianARD(F, tau, name='Y')
        X.initialize_from_value(x0[None, lo:hi])
        Y.observe(y[:, lo:hi], mask=mask[:, lo:hi])
        Q = VB(Y, F, W, X, tau, alpha)
        Q.ignore_bound_checks = True
        Q.update(repeat=2, verbose=False)
        res['L_before'] = Q.compute_lowerbound()
        R = transformations.RotationOptimizer(transformations.RotateGaussianARD(W, alpha),
                                              transformations.RotateGaussianARD(X), K)
        R.rotate()
        res['L_after'] = Q.compute_lowerbound()
        res['W_u0_rot'] = np.asarray(W.u[0])
        res['X_u0_rot'] = np.asarray(X.u[0])
        res['lo'], res['hi'] = lo, hi
    elif case == 'lssm':
        # batched linear state-space model (BASELINE config 5 shape): sequences sharded
        from bayespy_amd.nodes import GaussianARD, Gamma, SumMultiply, GaussianMarkovChain
        g = np.load(os.path.join(golden, 'lssm.npz'))
        tag = 'lssmB'
        y, x0, c0 = g[tag + '_y'], g[tag + '_x0'], g[tag + '_c0']
        M, B = y.shape[0], y.shape[1]
        T, D = x0.shape[-2], x0.shape[-1]
        lo, hi = B * rank // world, B * (rank + 1) // world
        alpha = Gamma(1e-5, 1e-5, plates=(D,), name='alpha')
        A = GaussianARD(0, alpha, shape=(D,), plates=(D,), name='A')
        A.initialize_from_value(np.identity(D))
        nu = Gamma(1e-3, 1e-3, plates=(D,), name='nu')
        X = GaussianMarkovChain(np.zeros(D), 1e-3 * np.identity(D), A, nu, n=T,
                                plates=(hi - lo,), name='X').shard(-1)
        X.initialize_from_value(x0[lo:hi])
        gamma = Gamma(1e-5, 1e-5, plates=(D,), name='gamma')
        gamma.initialize_from_value(1e-2 * np.ones(D))
        C = GaussianARD(0, gamma, shape=(D,), plates=(M, 1, 1), name='C')
        C.initialize_from_value(c0)
        tau = Gamma(1e-5, 1e-5, name='tau')
        tau.initialize_from_value(1e2)
        F = SumMultiply('i,i', C, X, name='F')          # inherits the partition from X
        Y = GaussianARD(F, tau, name='Y')
        Y.observe(y[:, lo:hi])
        Q = VB(Y, F, C, gamma, X, A, alpha, tau, nu)
        Q.ignore_bound_checks = True
        n = len(g[tag + '_L'])
        Q.update(repeat=n, verbose=False)
        res['L'] = np.array(Q.L[:n])
        for nm, nd in dict(A=A, C=C, tau=tau, alpha=alpha, gamma=gamma, nu=nu).items():
            res['L_' + nm] = np.array(Q.l[nd][:n])
        res['A_u0'] = np.asarray(A.u[0])
        res['engine'] = type(Q.plans[0]).__name__
    elif case in ('pca_fused_gram', 'pca_fused_stream', 'pca_generic'):
        # the headline block (both statistics forms) and the same model on the generic engine, the
        # plate split RAGGED over the two ranks (a third / two thirds)
        from models import build_pca
        from bayespy_amd.device import get_runtime
        g = np.load(os.path.join(golden, 'pca_n777_d20_k5.npz'))
        y, x0 = g['y'], g['x0']
        N = y.shape[1]
        cut = [0, N // 3, N]
        lo, hi = cut[rank], cut[rank + 1]
        kw = {'engine': 'generic'} if case == 'pca_generic' else {}
        Q = build_pca(nodes, VB, y[:, lo:hi], x0[lo:hi], x0.shape[1], shard=True, **kw)
        if case != 'pca_generic':
            Q.plans[0].stats = case.split('_')[-1]
        Q.ignore_bound_checks = True
        Q.update(repeat=int(g['n_iter']), verbose=False)
        res['engine'] = type(Q.plans[0]).__name__
        res['L'] = np.array(Q.L[:Q.iter])
        for nm in ('Y', 'X', 'W', 'tau', 'alpha'):
            res['L_' + nm] = np.array(Q.l[Q[nm]][:Q.iter])
        res['W_u0'], res['W_u1'] = np.asarray(Q['W'].u[0]), np.asarray(Q['W'].u[1])
        res['X_u0'] = np.asarray(Q['X'].u[0])
        res['tau_u0'], res['alpha_u1'] = np.asarray(Q['tau'].u[0]), np.asarray(Q['alpha'].u[1])
        rt = get_runtime()
        res['calls'] = np.array([rt.collective_calls['library'] + rt.collective_calls['torch']])
        res['lo'], res['hi'] = lo, hi
    elif case in ('gmm_fused', 'gmm_generic'):
        # the mixture block / the mixture on the generic engine, plate split ragged over the ranks
        from bayespy_amd.nodes import (GaussianARD, Gaussian, Wishart, Dirichlet, Categorical,
                                       Mixture)
        from bayespy_amd.device import get_runtime
        g = np.load(os.path.join(golden, 'gmm_n3000_d8_k16.npz'))
        y, lab0 = g['y'], g['lab0']
        N, D = y.shape
        K = g['alpha_u0'].shape[-1]
        cut = [0, N // 3 + 1, N]
        lo, hi = cut[rank], cut[rank + 1]
        alpha = Dirichlet(1e-3 * np.ones(K), name='alpha')
        z = Categorical(alpha, plates=(hi - lo,), name='z').shard(-1)
        mu = GaussianARD(0, 1e-3, shape=(D,), plates=(K,), name='mu')
        Lam = Wishart(D, 0.01 * np.identity(D), plates=(K,), name='Lambda')
        Y = Mixture(z, Gaussian, mu, Lam, plates=(hi - lo,), name='Y')
        z.initialize_from_value(lab0[lo:hi])
        Y.observe(y[lo:hi])
        Q = VB(Y, mu, Lam, z, alpha, engine='generic' if case == 'gmm_generic' else None)
        Q.ignore_bound_checks = True
        n = int(g['n_iter'])
        Q.update(repeat=n, verbose=False)
        res['engine'] = type(Q.plans[0]).__name__
        res['L'] = np.array(Q.L[:n])
        for nm in ('Y', 'mu', 'Lambda', 'z', 'alpha'):
            res['L_' + nm] = np.array(Q.l[Q[nm]][:n])
        res['mu_u0'], res['Lambda_u0'] = np.asarray(mu.u[0]), np.asarray(Lam.u[0])
        res['alpha_u0'], res['z_u0'] = np.asarray(alpha.u[0]), np.asarray(z.u[0])
        rt = get_runtime()
        res['calls'] = np.array([rt.collective_calls['library'] + rt.collective_calls['torch']])
        res['lo'], res['hi'] = lo, hi
    elif case == 'lssm_rotation':
        # the batch case of tests/golden/lssm_rotations.npz with the sequences split over the ranks:
        # the rotation statistics are global plate sums, every rank finds the same R and rotates
        # its own sequences
        import warnings
        from bayespy_amd.nodes import GaussianARD, Gamma, SumMultiply, GaussianMarkovChain
        g = np.load(os.path.join(golden, 'lssm_rotations.npz'))
        y, x0, c0 = g['batch_y'], g['batch_x0'], g['batch_c0']
        M, B = y.shape[0], y.shape[1]
        T, D = x0.shape[-2], x0.shape[-1]
        lo, hi = B * rank // world, B * (rank + 1) // world
        alpha = Gamma(1e-5, 1e-5, plates=(D,), name='alpha')
        A = GaussianARD(0, alpha, shape=(D,), plates=(D,), name='A')
        A.initialize_from_value(np.identity(D))
        X = GaussianMarkovChain(np.zeros(D), 1e-3 * np.identity(D), A, np.ones(D), n=T,
                                plates=(hi - lo,), name='X').shard(-1)
        X.initialize_from_value(x0[lo:hi])
        gamma = Gamma(1e-5, 1e-5, plates=(D,), name='gamma')
        gamma.initialize_from_value(1e-2 * np.ones(D))
        C = GaussianARD(0, gamma, shape=(D,), plates=(M, 1, 1), name='C')
        C.initialize_from_value(c0)
        tau = Gamma(1e-5, 1e-5, name='tau')
        tau.initialize_from_value(1e2)
        F = SumMultiply('i,i', C, X, name='F')
        Y = GaussianARD(F, tau, name='Y')
        Y.observe(y[:, lo:hi])
        Q = VB(Y, F, C, gamma, X, A, alpha, tau)
        Q.ignore_bound_checks = True
        res['engine'] = type(Q.plans[0]).__name__
        rotX = transformations.RotateGaussianMarkovChain(
            X, transformations.RotateGaussianARD(A, alpha, axis=0))
        R = transformations.RotationOptimizer(rotX, transformations.RotateGaussianARD(C, gamma, axis=0), D)
        Q.update(repeat=2, verbose=False)
        res['L_before'] = Q.compute_lowerbound()
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            R.rotate(maxiter=10)
        res['L_after'] = Q.compute_lowerbound()
        res['A_u0_rot'] = np.asarray(A.u[0])
        res['X_u0_rot'] = np.asarray(X.u[0])
        res['lo'], res['hi'] = lo, hi
    elif case == 'lssm_masked':
        # the state-space block with one mask per sequence (live-reference trace 'mb' of
        # lssm_masked.npz) on the DEVICE kernels, the sequences split 2 + 3 over the ranks: the set-up
        # counts and the raw plate sums of every X pass are all-reduced
        from test_lssm_masked_host import build
        g = np.load(os.path.join(golden, 'lssm_masked.npz'))
        y, mask, x0, c0 = g['mb_y'], g['mb_mask'], g['mb_x0'], g['mb_c0']
        B = y.shape[1]
        lo, hi = (0, 2) if rank == 0 else (2, B)
        Q, track = build(np.ascontiguousarray(y[:, lo:hi]), np.ascontiguousarray(mask[:, lo:hi]),
                         np.ascontiguousarray(x0[lo:hi]), c0, hi - lo, True, shard=True, host=False)
        res['engine'] = type(Q.plans[0]).__name__
        n = len(g['mb_L'])
        Q.update(repeat=n, verbose=False)
        res['L'] = np.array(Q.L[:n])
        res['C_u0'], res['A_u0'] = np.asarray(track['C'].u[0]), np.asarray(track['A'].u[0])
        res['X_u0'] = np.asarray(track['X'].u[0])
        res['lo'], res['hi'] = lo, hi
    elif case == 'hmm':
        # a batch of hidden Markov chains (case 3 of tests/models.py run_markov_chain_cases)
        # with the chain plate split over the ranks; emission and transition parameters are
        # replicated and receive all-reduced messages
        g = np.load(os.path.join(golden, 'markov_chains.npz'))
        yb, prior, z0 = g['in_hmm3_y'], g['in_hmm3_prior'], g['in_hmm3_z0']
        B, T = yb.shape
        K = prior.shape[-1]
        lo, hi = B * rank // world, B * (rank + 1) // world
        a0 = nodes.Dirichlet(np.ones(K), plates=(hi - lo,), name='a0').shard(-1)
        A = nodes.Dirichlet(prior, name='A')
        Z = nodes.CategoricalMarkovChain(a0, A, name='Z')
        m = nodes.GaussianARD(0, 1e-2, plates=(K,), name='m')
        t = nodes.Gamma(1e-1, 1e-1, plates=(K,), name='t')
        Y = nodes.Mixture(Z, nodes.GaussianARD, m, t, name='Y')
        Z.initialize_from_value(z0[lo:hi])
        Y.observe(yb[lo:hi])
        Q = VB(Y, m, t, Z, A, a0)
        Q.ignore_bound_checks = True
        Q.update(repeat=4, verbose=False)
        res['L'] = np.array(Q.L[:4])
        res['m_u0'], res['t_u0'] = np.asarray(m.u[0]), np.asarray(t.u[0])
        res['A_u0'] = np.asarray(A.u[0])
        res['Z_u0'] = np.asarray(Z.u[0])
        res['lo'], res['hi'] = lo, hi
    else:
        raise SystemExit('unknown case ' + case)
    np.savez(os.path.join(out, 'rank%d.npz' % rank), **res)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
