#!/usr/bin/env python
"""
bench.py -- VB iterations/sec of probabilistic PCA (BASELINE.json metric).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python bench.py --config {gmm,masked,lssm}        # the secondary BASELINE configurations
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    ... bench.py --gpus 8 --plates 100000000      # BASELINE config 4: N=1e8 over 8 ranks

One "step" = one full VB iteration exactly as ``VB.update`` does it
(vmp.py:154-172, :693-764): W, X, tau, alpha updated once in constructor order
plus the full lower bound (one device->host read of the ELBO per iteration).
Workload: BASELINE.json's metric config, PCA N=1e7, D=128, K=32, fully observed,
fp64, synthetic data of demos/pca.py:70-74 generated on the device.  With N
GPUs the observation plate N is sharded (strong scaling: the metric is quoted
on N=1e7 at 1/2/4/8 GPUs).  In the default Gram form the D x D Gram matrix of the data is
all-reduced once (RCCL) when the model is set up and an iteration needs no exchange at all;
in the streaming-statistics form (--stats stream) the child->parent message sums are one
all-reduce per iteration.

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement").
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X fp64 matrix peak (v_mfma_f64_16x16x4_f64: 32 FLOP/clk/SIMD
                               # x 1024 SIMDs x 2.4 GHz); not tabulated in the microarch guide


def parse():
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=20)
    p.add_argument('--warmup', type=int, default=3)
    # (--plates: the same under `python -m torch.distributed.run`, whose own parser rejects the
    # abbreviation-ambiguous `--n` even behind the script name)
    p.add_argument('--n', '--plates', dest='n', type=int, default=10_000_000,
                   help='total observation plate size')
    p.add_argument('--d', type=int, default=128)
    p.add_argument('--k', type=int, default=32)
    p.add_argument('--scaling', choices=['strong', 'weak'], default='strong')
    p.add_argument('--stats', choices=['gram', 'stream'], default='gram',
                   help="form of X.update()'s plate pass (see bayespy_amd/inference/plans/pca.py)")
    p.add_argument('--no-cpu-baseline', action='store_true')
    p.add_argument('--cpu-sample-n', type=int, default=0,
                   help='columns of the CPU-baseline sample; 0 = the whole workload when the '
                        'host has the memory for it (direct parity at the metric size)')
    p.add_argument('--config', choices=['pca', 'pca_c2', 'gmm', 'gmm_d16', 'masked', 'lssm',
                                        'lssm_d8', 'lssm_d16', 'lssm_masked', 'lssm_masked_1e5', 'generic_pca',
                                        'generic_gmm'],
                   default='pca',
                   help="pca = the BASELINE.json metric (default); the others print the "
                        "secondary configurations of tools/workloads.py as the JSON line")
    p.add_argument('--no-extra', action='store_true',
                   help='default run: do not append the secondary configurations under "extra"')
    p.add_argument('--layout', choices=['tiled', 'rows'], default=None)
    p.add_argument('--steady-steps', type=int, default=200,
                   help='a second, longer region after the timed one: per-step median / max '
                        '("steady" in the line); 0 to skip')
    p.add_argument('--exact-steps', action='store_true',
                   help='--config runs: time exactly --steps iterations (profiling runs); the '
                        'default raises short requests to the per-leg floor of >= 50 steps')
    p.add_argument('--full-out', default=None,
                   help='write the full (verbose) records of the headline and of every extra leg '
                        'to this JSON file; the printed line carries their compact forms')
    return p.parse_args()


def make_shard(torch, dev, n_local, D, K, seed, rank):
    """Synthetic PCA data of BASELINE.md section 3 for this rank's shard, generated
    on the device in chunks: y = w x + 0.1 eps, (D, n_local) with the plate contiguous."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    w = torch.randn(D, K, generator=g, device=dev, dtype=torch.float64)   # same w on all ranks
    g.manual_seed(seed + 1000 * (rank + 1))
    ld = (n_local + 31) // 32 * 32       # whole 32-column tiles (pad columns are zero)
    y = torch.empty(D, ld, device=dev, dtype=torch.float64)
    if ld != n_local:
        y[:, n_local:].zero_()
    step = 1 << 20
    for s in range(0, n_local, step):
        e = min(n_local, s + step)
        x = torch.randn(K, e - s, generator=g, device=dev, dtype=torch.float64)
        y[:, s:e] = w @ x
        y[:, s:e] += 0.1 * torch.randn(D, e - s, generator=g, device=dev, dtype=torch.float64)
    return y[:, :n_local]


def library_build_id():
    from tools import workloads
    return workloads.library_build_id()


def pmc_traffic(kernel, D, K, n_local):
    """HBM bytes per launch of the dominant kernel from a committed rocprofv3 PMC summary
    (profiles/r*/pmc_*.txt; separate FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for gfx950) -- ONLY if that profile was taken on this very
    build of the kernels (header line ``# build_id:`` written by tools/collect_profiles_r03.sh
    == vmp_version()) and on this workload (``# workload:``); otherwise (None, reason)."""
    from tools import workloads
    prof, why = workloads.pmc_profile('D=%d K=%d n_local=%d' % (D, K, n_local))
    if prof is None:
        return None, why
    for name, counters in prof['kernels'].items():
        if name.startswith(kernel):
            b = workloads.pmc_bytes(counters)
            if b is not None:
                return b, prof['path']
    return None, 'no %s counters in %s' % (kernel, prof['path'])


def cpu_baseline(y_dev, x0_dev, D, K, n_total, L_gpu, Q, sample_n):
    """The NumPy oracle (kind 'port') timed on this box's host cores on a bounded sample of the
    SAME workload -- by default the whole of it: the data and the injected initial <x> of this
    very run are pulled to the host, the oracle runs three iterations from them (two timed,
    ~10 s at N=1e7), and its lower bounds are compared with the first iterations of the run
    that was just timed (``elbo_rel_err_full``: direct parity at the metric size), its
    posterior means of W and of a strided sample of X with the device's."""
    import numpy as np
    import psutil
    from oracle.pca import PCAOracle
    try:
        from threadpoolctl import threadpool_info
        cores = max([i.get('num_threads', 1) for i in threadpool_info()] or [1])
    except Exception:       # noqa: BLE001
        cores = os.cpu_count() or 1
    need = 8.0 * n_total * (D + 3 * K) * 1.25
    full = sample_n <= 0 or sample_n >= n_total
    if full and psutil.virtual_memory().available < need:
        full, sample_n = False, 200_000
    n = n_total if full else min(sample_n, n_total)
    y = np.empty((D, n))
    step = max(1, (1 << 27) // D)
    for s0 in range(0, n, step):                       # bounded staging copies
        e = min(n, s0 + step)
        y[:, s0:e] = y_dev[:, s0:e].cpu().numpy()
    x0 = x0_dev[:n].cpu().numpy()
    iters = 3
    o = PCAOracle(y, x0, keep_x=True, chunk=1 << 17)
    o.iterate(1)
    t = time.time()
    o.iterate(iters - 1)
    dt = (time.time() - t) / (iters - 1)
    out = {'unit': 'VB iterations/s', 'cores': int(cores), 'kind': 'port'}
    if full:
        ncmp = min(iters, len(L_gpu))
        rel = max(abs(a - b) / abs(b) for a, b in zip(L_gpu[:ncmp], o.L[:ncmp]))
        out.update({
            'value': 1.0 / dt, 'elbo_rel_err_full': float(rel), 'elbo_iterations_compared': ncmp,
            'sample': 'oracle/pca.py on the WHOLE workload (data and initial <x> of this run), '
                      '%d timed iterations, %.2f s/iter' % (iters - 1, dt),
            'reference_note': 'unmodified reference: 35.4 s/iter at N=1e5 on 8 vCPU '
                              '(BASELINE.md section 2) = 2.8e-4 it/s at N=1e7'})
        return out
    # memory-bound host: the HIP path on the same sample, from the same initial moments
    from bayespy_amd import nodes
    from bayespy_amd.inference import VB
    alpha = nodes.Gamma(1e-2, 1e-2, plates=(K,))
    W = nodes.GaussianARD(0, alpha, shape=(K,), plates=(D, 1))
    X = nodes.GaussianARD(0, 1, shape=(K,), plates=(1, n))
    tau = nodes.Gamma(1e-2, 1e-2)
    Y = nodes.GaussianARD(nodes.SumMultiply('i,i', W, X), tau)
    X.initialize_from_value(x0[None])
    Y.observe(y)
    Qs = VB(Y, W, X, tau, alpha)
    Qs.ignore_bound_checks = True
    Qs.update(repeat=iters, verbose=False)
    rel = max(abs(a - b) / abs(b) for a, b in zip(Qs.L[:iters], o.L))
    out.update({
        'value': 1.0 / (dt * (n_total / float(n))), 'elbo_rel_err_hip_vs_oracle': float(rel),
        'elbo_iterations_compared': iters,
        'sample': 'oracle/pca.py (NumPy fp64, BLAS GEMMs) on the first N=%d columns of the same '
                  'D=%d,K=%d data, %d timed iterations at %.3f s/iter, extrapolated linearly to '
                  'N=%d (the host lacks the memory for the whole workload)'
                  % (n, D, K, iters - 1, dt, n_total)})
    return out


def run_extra(name, fn, budget_s, **kw):
    """One secondary configuration for the "extra" list of the default run; a failure there is
    reported, never fatal to the headline line."""
    import torch
    t = time.time()
    torch.cuda.reset_peak_memory_stats()        # peak_mem_GB of a leg is that leg's own
    try:
        out = fn(**kw)
    except Exception as e:       # noqa: BLE001
        out = {'metric': name, 'error': '%s: %s' % (type(e).__name__, e)}
    out['wall_s'] = time.time() - t
    torch.cuda.empty_cache()
    return out


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    # launched by torch.distributed.run (RANK is set) -- also with ONE rank, so that a world-1
    # run exercises the library's RCCL communicator exactly as the N-rank runs do
    launched = 'RANK' in os.environ and 'MASTER_PORT' in os.environ
    if world > 1 or launched:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        ndev = torch.cuda.device_count()
        torch.cuda.set_device(local_rank % ndev)
        # "nccl" is RCCL on ROCm.  VMP_BENCH_BACKEND=gloo exists only to smoke-test this launch
        # path with several ranks on a one-GPU box (RCCL refuses two ranks on one device).
        backend = os.environ.get('VMP_BENCH_BACKEND', 'nccl')
        if backend == 'nccl':
            # the plate sums of a benchmarked run must be the library's own RCCL all-reduce
            # (vmp_allreduce_sum_f64): a fallback to torch.distributed.all_reduce is an ERROR
            # here, not a warning (bayespy_amd/device.py)
            os.environ.setdefault('BAYESPY_AMD_COLLECTIVE', 'library')
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank % ndev))
        else:
            dist.init_process_group(backend)
    if args.gpus != world and rank == 0 and world > 1:
        print('warning: --gpus %d but WORLD_SIZE %d' % (args.gpus, world), file=sys.stderr)

    from bayespy_amd import nodes
    from bayespy_amd.inference import VB
    from bayespy_amd.device import get_runtime

    rt = get_runtime()
    dev = rt.device
    if args.config != 'pca':
        # the secondary BASELINE configurations share the JSON contract (tools/workloads.py)
        from tools import workloads
        # every leg times >= 50 steps (a 10-step region of a 2 ms step is one hiccup away from
        # any number: VERDICT r03 weak #1)
        table = {
            'pca_c2': ('run_pca_c2', dict(), 500),
            'gmm': ('run_gmm', dict(), 100),
            'gmm_d16': ('run_gmm', dict(N=4_000_000, D=16, K=32, cpu_sample_n=50_000), 100),
            'masked': ('run_masked', dict(), 50),
            'lssm': ('run_lssm', dict(), 100),
            'lssm_d8': ('run_lssm', dict(D=8, cpu_sample_b=4000), 50),
            'lssm_d16': ('run_lssm', dict(D=16, cpu_sample_b=2000), 50),
            'lssm_masked': ('run_lssm_masked', dict(), 50),
            'lssm_masked_1e5': ('run_lssm_masked', dict(B=100_000), 20),
            'generic_pca': ('run_generic_pca', dict(), 50),
            'generic_gmm': ('run_generic_gmm', dict(), 50),
        }
        fname, kw, floor = table[args.config]
        if args.exact_steps:
            floor = 1
        out = getattr(workloads, fname)(steps=max(args.steps, floor),
                                        warmup=args.warmup if args.exact_steps else max(args.warmup, 2),
                                        cpu_baseline=not args.no_cpu_baseline, **kw)
        if rank == 0:
            out['comm'] = rt.comm_info()
            print(json.dumps(out))
        elif dist.is_initialized():
            rt.comm_info()          # collective inside (_ensure_comm): every rank takes part
        if dist.is_initialized():
            dist.destroy_process_group()
        return
    D, K = args.d, args.k
    if args.scaling == 'strong':
        n_total = args.n
        lo = n_total * rank // world
        hi = n_total * (rank + 1) // world
        n_local = hi - lo
    else:
        n_local = args.n
        n_total = args.n * world

    y = make_shard(torch, dev, n_local, D, K, seed=42, rank=rank)

    # ---- the model, written exactly like a BayesPy script (demos/pca.py:22-61) ----
    alpha = nodes.Gamma(1e-2, 1e-2, plates=(K,), name='alpha')
    W = nodes.GaussianARD(0, alpha, shape=(K,), plates=(D, 1), name='W')
    X = nodes.GaussianARD(0, 1, shape=(K,), plates=(1, n_local), name='X')
    if world > 1 or launched:
        # the observation plate is partitioned over the ranks.  Declared for a
        # world-1 launch under torch.distributed.run too: the set-up plate sums (G, sum y^2) then
        # go through vmp_allreduce_sum_f64 on a one-rank RCCL communicator, i.e. the collective
        # path of the N-rank runs executes on every driver run (comm.calls.library > 0)
        X.shard(-1)
    F = nodes.SumMultiply('i,i', W, X, name='F')
    tau = nodes.Gamma(1e-2, 1e-2, name='tau')
    Y = nodes.GaussianARD(F, tau, name='Y')
    # the initial <x> are injected draws (initialize_from_value), the same
    # arrays the CPU baseline starts from
    gx = torch.Generator(device=dev)
    gx.manual_seed(4242 + rank)
    x0 = torch.randn(n_local, K, generator=gx, device=dev, dtype=torch.float64)
    X.initialize_from_value(x0[None])
    Y.observe(y)
    Q = VB(Y, F, W, X, tau, alpha)
    Q.ignore_bound_checks = True          # never stop early: time exactly K iterations
    plan = Q.plans[0]
    plan.stats = args.stats
    if args.layout:
        plan.plate_layout = args.layout

    # set-up: device state, tile-major copy of Y, the placement trial of the plate arrays -- here
    # rather than inside the first iteration, so that --warmup 0 times only iterations.  Its cost
    # (wall time, transient device memory) is reported in config.setup
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    mem_before = torch.cuda.memory_allocated()
    t_setup = time.perf_counter()
    if hasattr(plan, 'place_plate_arrays') and args.stats == 'gram':
        plan.place_plate_arrays()
    torch.cuda.synchronize()
    setup = {'setup_s': time.perf_counter() - t_setup,
             'data_GB': mem_before / 1e9,
             'setup_peak_GB': torch.cuda.max_memory_allocated() / 1e9}
    setup['resident_GB'] = torch.cuda.memory_allocated() / 1e9
    setup['reserved_after_GB'] = torch.cuda.memory_reserved() / 1e9

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    Q.update(repeat=args.warmup, verbose=False)
    plan.enable_timing(True)              # HIP events around the pass kernel, on its stream
    barrier()
    t0 = time.perf_counter()
    Q.update(repeat=args.steps, verbose=False)
    barrier()
    dt = time.perf_counter() - t0
    # one HIP event triple per pass was recorded inside the timed region (ring of 64);
    # read them only now so that no iteration blocks on the host
    pass_ms = plan.pass_times_ms(64)
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    L = Q.L[:Q.iter]
    steady = None
    if args.steady_steps > 0:
        # the contract's region above is K steps (20 x 2.2 ms on the driver's command): a second,
        # longer region of the same loop with the per-iteration wall times VB records
        # (Q.cputime) -- mean, median, max
        from tools import workloads
        _, steady = workloads.timed_update(Q, max(args.steady_steps, args.steps), barrier)
        steady['steps'] = max(args.steady_steps, args.steps)
        sp = plan.pass_times_ms(64)
        steady['pass_ms_mean'] = sum(p[0] for p in sp) / len(sp)
        steady['pass_ms_min'] = min(p[0] for p in sp)
        steady['pass_ms_max'] = max(p[0] for p in sp)
    comm = rt.comm_info()          # every rank: the first call may create the communicator

    if rank == 0:
        ms_step = 1e3 * dt / args.steps
        it_s = args.steps / dt
        avg_pass = sum(p[0] for p in pass_ms) / len(pass_ms)
        avg_red = sum(p[1] for p in pass_ms) / len(pass_ms)
        # algorithmic work of ONE launch of the dominant kernel (pca_pass_kernel) on this
        # rank: SURVEY.md 8(d): bytes = 8 N (D+K), flops = 4 N D K + 2 N K^2
        alg_bytes = 8.0 * n_local * (D + K)
        alg_flops = 4.0 * n_local * D * K + 2.0 * n_local * K * K
        if args.stats == 'gram':
            # Gram form: the plate kernel only computes X = A Y (2 N D K flops)
            alg_flops = 2.0 * n_local * D * K
        tflops = alg_flops / (avg_pass * 1e-3) / 1e12
        gbs = alg_bytes / (avg_pass * 1e-3) / 1e9
        if args.stats == 'gram':
            roof = {'kernel': 'pca_xpass_kernel', 'bound': 'hbm', 'achieved': gbs,
                    'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': gbs / HBM_PEAK_GBS,
                    'traffic': None, 'avg_launch_ms': avg_pass, 'gram_stats_ms': avg_red,
                    'mfma_TFLOPs': tflops, 'mfma_frac_of_78.6': tflops / FP64_MFMA_PEAK_TFLOPS,
                    'alg_bytes_per_launch': alg_bytes, 'alg_flops_per_launch': alg_flops}
        else:
            roof = {'kernel': 'pca_pass_kernel', 'bound': 'mfma', 'achieved': tflops,
                    'peak': FP64_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                    'frac': tflops / FP64_MFMA_PEAK_TFLOPS, 'traffic': None,
                    'avg_launch_ms': avg_pass, 'reduce_ms': avg_red, 'hbm_achieved_GBs': gbs,
                    'hbm_frac_of_8TBs': gbs / HBM_PEAK_GBS,
                    'alg_bytes_per_launch': alg_bytes, 'alg_flops_per_launch': alg_flops}
        out = {
            'metric': 'VB iterations/sec, PCA N=%d D=%d K=%d' % (n_total, D, K),
            'value': it_s, 'unit': 'VB iterations/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': ms_step, 'higher_is_better': True,
            'scaling': args.scaling, 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {
                'workload': 'PCA (GaussianARD+SumMultiply+Gamma) N=%d D=%d K=%d, fully observed; '
                            'step = W,X,tau,alpha updates + full ELBO' % (n_total, D, K),
                'n_local': n_local, 'parallelism': 'plate-shard x%d' % world,
            },
            'elbo_first': float(L[0]), 'elbo_last': float(L[-1]),
            'roofline': roof,
        }
        roof['traffic'], roof['traffic_source'] = pmc_traffic(roof['kernel'], D, K, n_local)
        roof['library_build'] = library_build_id()
        out['config']['stats'] = args.stats
        out['config']['plate_layout'] = plan.plate_layout if args.stats == 'gram' else 'rows'
        out['config']['initial_x'] = 'initialize_from_value (injected normal draws)'
        # set-up: allocations of X / tile-major Y tried, pass time on each (plans/pca.py)
        pl = getattr(plan, 'placement', None)
        if pl:
            flat = [v for row in pl['grid_ms'] for v in row]
            setup['placement'] = {'candidates': [len(pl['yt_ms']) - 1, len(pl['grid_ms'][-1])],
                                  'kept_ms': round(min(flat), 4), 'first_ms': round(flat[0], 4),
                                  'worst_ms': round(max(flat), 4)}
            # what a user gets who cannot afford the trial (BAYESPY_AMD_PLACEMENT_TRIES=1: the
            # first allocation is kept): this run's iteration with its pass replaced by the pass
            # time measured on the first allocation pair -- the iteration is pass-bound, the
            # replicated-node chain runs beside the pass
            first_step = ms_step - avg_pass + flat[0]
            out['value_first_allocation'] = {
                'value': 1e3 / first_step, 'ms_per_step': first_step,
                'roofline_frac': alg_bytes / (flat[0] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                'note': 'estimate: ms_per_step - kept pass + first-allocation pass (both measured '
                        'in this run); `value` above is with the placement trial'}
        else:
            setup['placement'] = None
        out['config']['setup'] = {k: (round(v, 3) if isinstance(v, float) else v)
                                  for k, v in setup.items()}
        out['peak_mem_GB'] = round(torch.cuda.max_memory_allocated() / 1e9, 2)
        if steady is not None:
            out['steady'] = {k: (round(v, 4) if isinstance(v, float) else v)
                             for k, v in steady.items()}
        out['comm'] = comm
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(y, x0, D, K, n_total, [float(v) for v in L], Q,
                                               args.cpu_sample_n)
        full = {'headline': dict(out, placement_grid=pl)}
        if world == 1 and not args.no_extra and (n_total, D, K) == (10_000_000, 128, 32):
            # the other BASELINE configurations that fit one GPU, on the driver-visible line: ONE
            # compact record per leg (tools/workloads.py:compact -- step time with median / max,
            # dominant kernel time, roofline fraction, in-run parity, CPU baseline), LAST in the
            # line so that a tail of it holds all of them; the verbose records go to --full-out.
            # Every leg times >= 50 steps or >= 0.5 s
            del Q, plan, Y, F, W, X, tau, alpha, y, x0
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            from tools import workloads
            cb = not args.no_cpu_baseline
            legs = [
                ('pca_c2', 'run_pca_c2', dict(steps=2000, warmup=20)),
                ('gmm', 'run_gmm', dict(steps=200, warmup=5)),
                ('gmm_d16', 'run_gmm', dict(N=4_000_000, D=16, K=32, steps=300, warmup=5,
                                            cpu_sample_n=50_000)),
                ('masked', 'run_masked', dict(steps=50, warmup=1)),
                ('lssm', 'run_lssm', dict(steps=150, warmup=3)),
                ('lssm_d8', 'run_lssm', dict(D=8, steps=50, warmup=2, cpu_sample_b=4000)),
                ('lssm_d16', 'run_lssm', dict(D=16, steps=50, warmup=2, cpu_sample_b=2000)),
                ('lssm_masked', 'run_lssm_masked', dict(steps=50, warmup=2)),
                ('lssm_masked_1e5', 'run_lssm_masked', dict(B=100_000, steps=20, warmup=2)),
                ('generic_pca', 'run_generic_pca', dict(steps=50, warmup=4)),
                ('generic_gmm', 'run_generic_gmm', dict(steps=50, warmup=4)),
            ]
            out['extra'] = []
            for name, fname, kw in legs:
                fn = getattr(workloads, fname, None)
                if fn is None:
                    continue
                rec = run_extra(name, fn, 120, cpu_baseline=cb, **kw)
                full[name] = rec
                out['extra'].append(workloads.compact(rec, name))
        if args.full_out:
            with open(args.full_out, 'w') as f:
                json.dump(full, f, indent=1)
        print(json.dumps(out))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
