"""
The secondary BASELINE.json configurations as functions that return the bench.py JSON fields
(metric / value / ms_per_step / roofline / cpu_baseline):

    run_pca_c2  config 2  probabilistic PCA N=1e6, D=64, K=16 (launch / latency-bound: SURVEY.md 8d)
    run_gmm     config 3  Gaussian mixture N=1e7, D=8, K=64
    run_masked  SURVEY.md 8(d) secondary run: PCA N=1e7, D=128, K=32 with 10 % missing values
    run_lssm    config 5  linear state-space model, T=1e3 x B=1e5 sequences

``bench.py --config {pca_c2,gmm,masked,lssm}`` prints one of them as its JSON line; the default
``bench.py`` run (the PCA headline) appends all of them under "extra".  tools/bench_*.py are the
stand-alone command lines of the same functions.
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6
HBM_PEAK_GBS = 8000.0


def _cores():
    try:
        from threadpoolctl import threadpool_info
        return int(max([i.get('num_threads', 1) for i in threadpool_info()] or [1]))
    except Exception:       # noqa: BLE001
        return os.cpu_count() or 1


def library_build_id():
    from bayespy_amd import _lib
    v = _lib.load().vmp_version().decode()
    return v.split('build ')[-1] if 'build ' in v else None


def pmc_profile(workload):
    """The committed rocprofv3 PMC summary (profiles/r*/pmc_*.txt, written by
    tools/collect_profiles_r03.sh) of THIS build of the kernels (``# build_id:`` ==
    vmp_version()) on THIS workload (``# workload:``), parsed into
    ``{kernel: {counter: (avg, n)}}`` plus the header -- or (None, reason).  Never a replay of
    another build's counters."""
    import glob
    import re
    bid = library_build_id()
    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*', 'pmc_*.txt')), reverse=True):
        head, per, cur = {}, {}, None
        for line in open(path):
            m = re.match(r'#\s*(\w+):\s*(.+)', line)
            if m:
                head[m.group(1)] = m.group(2).strip()
                continue
            if line.startswith('=='):
                continue
            if not line.startswith(' '):
                cur = line.strip()
                continue
            m = re.match(r'\s+(\w+)\s+avg\s+([0-9.eE+-]+)\s+\(n=(\d+)', line)
            if m and cur:
                per.setdefault(cur, {})[m.group(1)] = (float(m.group(2)), int(m.group(3)))
        if head.get('build_id') == bid and head.get('workload') == workload:
            return {'kernels': per, 'head': head, 'path': os.path.relpath(path, ROOT)}, None
    return None, 'no committed PMC profile of build %s for "%s"' % (bid, workload)


def pmc_bytes(counters):
    """HBM bytes of one launch from the FETCH_SIZE / WRITE_SIZE averages (KB; FETCH_SIZE counts
    half of a wide streaming read on gfx950: MI355X_MICROARCH.md, HBM section)."""
    if 'FETCH_SIZE' not in counters or 'WRITE_SIZE' not in counters:
        return None
    return 2.0 * counters['FETCH_SIZE'][0] * 1024 + counters['WRITE_SIZE'][0] * 1024


def pmc_iteration_traffic(prof, prefix):
    """Sum over the kernels whose name starts with ``prefix``: bytes per launch x launches,
    divided by the iterations the profiled command ran (``# iterations:``).  The set-up kernels
    (relayout / prepare: run once) are excluded by name."""
    its = float(prof['head'].get('iterations', 0) or 0)
    if its <= 0:
        return None
    tot = 0.0
    for k, c in prof['kernels'].items():
        if not k.startswith(prefix) or any(w in k for w in ('prepare', 'relayout', 'x_layout',
                                                            'init')):
            continue
        b = pmc_bytes(c)
        if b is None:
            return None
        tot += b * c['FETCH_SIZE'][1]
    return tot / its


def timed_update(Q, steps, barrier=None):
    """``Q.update(repeat=steps)`` as ONE timed region (the contract of bench.py) plus the
    per-iteration wall times VB records itself (``Q.cputime``: node updates + the lower bound's
    device -> host read, vmp.py:712-717): mean over the region, median, max and where the max fell
    -- a single stall (a first-touch allocation, a host hiccup) shows up as ``max`` at a step
    index instead of silently moving a short region's mean."""
    import numpy as np
    import torch
    sync = barrier or torch.cuda.synchronize
    i0 = Q.iter
    sync()
    t0 = time.perf_counter()
    Q.update(repeat=steps, verbose=False)
    sync()
    dt = time.perf_counter() - t0
    per = 1e3 * np.asarray(Q.cputime[i0:i0 + steps], dtype=np.float64)
    st = {'ms_mean': 1e3 * dt / steps, 'ms_median': float(np.median(per)),
          'ms_max': float(per.max()), 'argmax_step': int(per.argmax()), 'timed_s': dt}
    return dt, st


def compact(rec, leg):
    """The one-screen form of a leg's record for the "extra" list of the default bench line (the
    driver keeps only the tail of that line): step time with its spread, the dominant kernel's
    time and roofline fraction, in-run parity, CPU baseline.  No prose."""
    if 'error' in rec:
        return {'leg': leg, 'error': str(rec['error'])[:160], 'wall_s': round(rec.get('wall_s', 0), 1)}
    roof = rec.get('roofline', {})
    cpu = rec.get('cpu_baseline', {})
    sp = rec.get('step_ms', {})

    def r(v, n=4):
        return None if v is None else float('%.*g' % (n, v))
    out = {'leg': leg, 'steps': rec.get('steps'), 'ms_per_step': r(rec.get('ms_per_step')),
           'ms_median': r(sp.get('ms_median')), 'ms_max': r(sp.get('ms_max')),
           'it_s': r(rec.get('value')),
           'bound': roof.get('bound'), 'frac': r(roof.get('frac'), 3),
           'kernel_ms': r(roof.get('avg_launch_ms'))}
    if sp.get('ms_max') and sp.get('ms_median') and sp['ms_max'] > 1.25 * sp['ms_median']:
        out['argmax_step'] = sp.get('argmax_step')      # where the one slow step fell
    if roof.get('frac_alg') is not None:
        out['frac_alg'] = r(roof['frac_alg'], 3)
    if roof.get('kernel_ms'):
        out['kernel_ms'] = {k: r(v) for k, v in roof['kernel_ms'].items()
                            if isinstance(v, (int, float))}
    if roof.get('issued_mfma_TFLOPs') is not None:
        out['issued_TFLOPs'] = r(roof['issued_mfma_TFLOPs'], 3)
    if roof.get('traffic') is not None:
        alg = roof.get('alg_bytes_per_launch') or roof.get('alg_bytes_per_iteration')
        out['traffic_GB'] = r(roof['traffic'] / 1e9)
        if alg:
            out['traffic_over_alg'] = r(roof['traffic'] / alg, 3)
    par = cpu.get('elbo_rel_err_full', cpu.get('elbo_rel_err_hip_vs_oracle'))
    if par is not None:
        out['elbo_rel_err'] = r(par, 2)
        out['parity_on'] = 'whole' if 'elbo_rel_err_full' in cpu else 'sample'
    if cpu.get('value') is not None:
        out['cpu_it_s'] = r(cpu['value'], 3)
        if cpu.get('cores') not in (None, 128):       # (128 = all cores of the box: the default)
            out['cpu_cores'] = cpu.get('cores')
    if rec.get('peak_mem_GB') is not None:
        out['peak_GB'] = r(rec['peak_mem_GB'], 3)
    sg = rec.get('config', {}).get('sweep_graph')
    if sg is not None:
        out['sweep'] = 'graph' if sg.get('recorded') else 'eager'
    # (the wall time of a leg is in the verbose record, --full-out: nine legs must fit a 5 KB line)
    return out


def run_pca_c2(N=1_000_000, D=64, K=16, steps=50, warmup=5, cpu_baseline=True):
    """BASELINE config 2: the headline model at a size where the replicated-node chain, not the
    plate pass, sets the step (bytes 0.64 GB = 0.08 ms at 8 TB/s): absolute it/s is the figure."""
    import numpy as np
    import torch
    from bayespy_amd.nodes import GaussianARD, Gamma, SumMultiply
    from bayespy_amd.inference import VB
    dev = torch.device('cuda', torch.cuda.current_device())
    g = torch.Generator(device=dev)
    g.manual_seed(42)
    w = torch.randn(D, K, generator=g, device=dev, dtype=torch.float64)
    x = torch.randn(K, N, generator=g, device=dev, dtype=torch.float64)
    y = w @ x + 0.1 * torch.randn(D, N, generator=g, device=dev, dtype=torch.float64)
    x0 = torch.randn(N, K, generator=g, device=dev, dtype=torch.float64)
    del w, x
    alpha = Gamma(1e-2, 1e-2, plates=(K,), name='alpha')
    W = GaussianARD(0, alpha, shape=(K,), plates=(D, 1), name='W')
    X = GaussianARD(0, 1, shape=(K,), plates=(1, N), name='X')
    F = SumMultiply('i,i', W, X, name='F')
    tau = Gamma(1e-2, 1e-2, name='tau')
    Y = GaussianARD(F, tau, name='Y')
    X.initialize_from_value(x0[None])
    Y.observe(y)
    Q = VB(Y, F, W, X, tau, alpha)
    Q.ignore_bound_checks = True
    plan = Q.plans[0]
    Q.update(repeat=warmup, verbose=False)
    # the step time of this leg is measured WITHOUT the per-pass HIP event triples: at this size three
    # hipEventRecord per iteration on the plate stream are ~25 us of queue serialisation (0.146 ms with,
    # 0.120 ms without, same process: tools/c2_fuse_ab.py, profiles/r06/c2_fuse_ab.txt) -- which is the
    # "slip" of this leg between rounds 4 and 5.  The pass kernel's own duration comes from a second
    # region of the same length with the events on (reported beside it).
    dt, step_ms = timed_update(Q, steps)
    plan.enable_timing(True)
    dt_ev, _ = timed_update(Q, min(steps, 60))
    pass_ms = plan.pass_times_ms(64)
    plan.enable_timing(False)
    avg_pass = sum(p[0] for p in pass_ms) / len(pass_ms)
    alg_bytes = 8.0 * N * (D + K)
    gbs = alg_bytes / (avg_pass * 1e-3) / 1e9
    L = [float(v) for v in Q.L[:Q.iter]]
    out = {
        'metric': 'VB iterations/sec, PCA N=%d D=%d K=%d' % (N, D, K),
        'value': steps / dt, 'unit': 'VB iterations/s', 'n_gpus': 1, 'steps': steps,
        'warmup': warmup, 'ms_per_step': 1e3 * dt / steps, 'higher_is_better': True,
        'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': 'probabilistic PCA (BASELINE config 2), N=%d D=%d K=%d, fully observed'
                               % (N, D, K), 'stats': plan.stats, 'plate_layout': plan.plate_layout},
        'elbo_first': L[0], 'elbo_last': L[-1], 'step_ms': step_ms,
        'roofline': {'kernel': 'pca_xpass_kernel', 'bound': 'hbm', 'achieved': gbs,
                     'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': gbs / HBM_PEAK_GBS,
                     'traffic': None, 'avg_launch_ms': avg_pass,
                     'alg_bytes_per_launch': alg_bytes,
                     'ms_per_step_with_pass_events': 1e3 * dt_ev / min(steps, 60),
                     'note': 'the step (ms_per_step, timed without per-pass events) is set by the '
                             'replicated-node chain at this size, not by this kernel; '
                             'avg_launch_ms from a second region with HIP events on the plate stream'},
    }
    if cpu_baseline:
        from oracle.pca import PCAOracle
        yh, xh = y.cpu().numpy(), x0.cpu().numpy()
        o = PCAOracle(yh, xh, keep_x=False)
        n_it = min(3, len(L))
        t0 = time.perf_counter()
        o.iterate(n_it)
        dtc = time.perf_counter() - t0
        out['cpu_baseline'] = {
            'value': n_it / dtc, 'unit': 'VB iterations/s', 'cores': _cores(), 'kind': 'port',
            'elbo_rel_err_full': float(np.max(np.abs((np.array(o.L) - np.array(L[:n_it]))
                                                     / np.array(o.L)))),
            'sample': 'oracle/pca.py on the whole workload, %d iterations at %.2f s/iter'
                      % (n_it, dtc / n_it)}
    return out


def run_gmm(N=10_000_000, D=8, K=64, steps=10, warmup=2, cpu_baseline=True, cpu_sample_n=100_000):
    import numpy as np
    import torch
    from bayespy_amd.nodes import GaussianARD, Gaussian, Wishart, Dirichlet, Categorical, Mixture
    from bayespy_amd.inference import VB
    dev = torch.device('cuda', torch.cuda.current_device())
    g = torch.Generator(device=dev)
    g.manual_seed(42)
    centers = 3 * torch.randn(K, D, generator=g, device=dev, dtype=torch.float64)
    lab = torch.randint(0, K, (N,), generator=g, device=dev)
    y = centers[lab] + 0.5 * torch.randn(N, D, generator=g, device=dev, dtype=torch.float64)
    lab0 = torch.randint(0, K, (N,), generator=g, device=dev)
    alpha = Dirichlet(1e-3 * np.ones(K), name='alpha')
    z = Categorical(alpha, plates=(N,), name='z')
    mu = GaussianARD(0, 1e-3, shape=(D,), plates=(K,), name='mu')
    Lam = Wishart(D, 0.01 * np.identity(D), plates=(K,), name='Lambda')
    Y = Mixture(z, Gaussian, mu, Lam, plates=(N,), name='Y')
    z.initialize_from_value(lab0.cpu().numpy())       # one-hot initial moments
    Y.observe(y)
    Q = VB(Y, mu, Lam, z, alpha)
    Q.ignore_bound_checks = True
    plan = Q.plans[0]
    Q.update(repeat=warmup, verbose=False)
    plan.enable_timing(True)
    dt, step_ms = timed_update(Q, steps)
    ms = plan.pass_times_ms(64)                    # HIP events recorded inside the timed region
    avg = sum(m[0] for m in ms) / len(ms)
    FS = D * D + D + 1
    flops = 4.0 * N * K * FS                       # 4 N K (D^2 + D + 1)
    byts = 8.0 * N * (D + K)
    tflops = flops / (avg * 1e-3) / 1e12
    out = {
        'metric': 'VB iterations/sec, GMM N=%d D=%d K=%d' % (N, D, K), 'value': steps / dt,
        'unit': 'VB iterations/s', 'n_gpus': 1, 'steps': steps, 'warmup': warmup,
        'ms_per_step': 1e3 * dt / steps, 'higher_is_better': True, 'scaling': 'strong',
        'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': 'Gaussian mixture (Mixture+Categorical+GaussianARD+Wishart+'
                               'Dirichlet), N=%d D=%d K=%d, one VB iteration = mu, Lambda, z, '
                               'alpha updates + full ELBO' % (N, D, K)},
        'elbo_first': float(Q.L[0]), 'elbo_last': float(Q.L[Q.iter - 1]), 'step_ms': step_ms,
        'roofline': {'kernel': 'gmm_pass_kernel', 'bound': 'mfma', 'achieved': tflops,
                     'peak': FP64_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                     'frac': tflops / FP64_MFMA_PEAK_TFLOPS, 'traffic': None,
                     'avg_launch_ms': avg, 'reduce_ms': sum(m[1] for m in ms) / len(ms),
                     'hbm_achieved_GBs': byts / (avg * 1e-3) / 1e9,
                     'alg_flops_per_launch': flops, 'alg_bytes_per_launch': byts},
    }
    # what the matrix unit actually executes: per 16-point tile KP/16 x F2P/4 (phase 1) +
    # 4 x KP/16 x F2P/16 (phase 2) instructions of 2048 flops over the compact feature list
    # (F2 = D(D+1)/2 + D + 1 padded to 16) -- about 2/3 of the algorithmic count, which prices the
    # symmetric quadratic form as D^2 products
    F2P, KP = int(plan.layout.F2P), int(plan.layout.KP)      # padded sizes the kernels run on
    issued = (N / 16.0) * (KP // 16) * (F2P // 4 + 4 * (F2P // 16)) * 2048.0
    out['roofline']['issued_mfma_flops_per_launch'] = issued
    out['roofline']['issued_mfma_TFLOPs'] = issued / (avg * 1e-3) / 1e12
    # frac = what the matrix unit EXECUTES over its peak; the SURVEY-formula
    # figure, which also counts the symmetric half the kernel rightly skips, is an efficiency
    out['roofline']['frac_alg'] = out['roofline']['frac']
    out['roofline']['achieved_alg'] = out['roofline']['achieved']
    out['roofline']['achieved'] = out['roofline']['issued_mfma_TFLOPs']
    out['roofline']['frac'] = out['roofline']['issued_mfma_TFLOPs'] / FP64_MFMA_PEAK_TFLOPS
    out['roofline']['note'] = ('achieved / frac: flops the matrix unit EXECUTES (instruction count x '
                               '2048, = SQ_INSTS_MFMA of the committed counters) over the pass time; '
                               'achieved_alg / frac_alg: the ALGORITHMIC flops of SURVEY.md 8(d), an '
                               'efficiency figure (ceiling of v_mfma_f64_16x16x4_f64 on this chip: '
                               '44-47 TFLOP/s, tools/mfma4_lab.hip)')
    prof, why = pmc_profile('GMM N=%d D=%d K=%d' % (N, D, K))
    if prof is not None:
        for name, counters in prof['kernels'].items():
            if name.startswith('gmm_pass_kernel') and pmc_bytes(counters) is not None:
                out['roofline']['traffic'] = pmc_bytes(counters)
        out['roofline']['traffic_source'] = prof['path']
    else:
        out['roofline']['traffic_source'] = why
    if cpu_baseline:
        from oracle.gmm import GMMOracle
        ns = min(cpu_sample_n, N)
        # the FIRST ns rows of the very data of this run, same initial labels: the lower bound of
        # the HIP path on that sample against the oracle's, and the oracle's time per iteration
        ys, l0 = y[:ns].cpu().numpy(), lab0[:ns].cpu().numpy()
        o = GMMOracle(ys, l0, K)
        o.iterate(1, keep_r=False)
        t = time.time()
        o.iterate(2, keep_r=False)
        dtc = (time.time() - t) / 2
        zs = Categorical(Dirichlet(1e-3 * np.ones(K)), plates=(ns,))
        mus = GaussianARD(0, 1e-3, shape=(D,), plates=(K,))
        Ls = Wishart(D, 0.01 * np.identity(D), plates=(K,))
        Ysm = Mixture(zs, Gaussian, mus, Ls, plates=(ns,))
        zs.initialize_from_value(l0)
        Ysm.observe(ys)
        Qs = VB(Ysm, mus, Ls, zs, zs.parents[0])
        Qs.ignore_bound_checks = True
        Qs.update(repeat=3, verbose=False)
        rel = max(abs(a - b) / abs(b) for a, b in zip(Qs.L[:3], o.L))
        out['cpu_baseline'] = {
            'value': 1.0 / (dtc * (N / float(ns))), 'unit': 'VB iterations/s', 'cores': _cores(),
            'kind': 'port', 'elbo_rel_err_hip_vs_oracle': float(rel),
            'sample': 'oracle/gmm.py (NumPy fp64, chunked) on the first N=%d rows of the same '
                      'data, 2 timed iterations at %.3f s/iter, extrapolated linearly to N=%d'
                      % (ns, dtc, N)}
    return out


def run_generic_pca(N=1_000_000, D=64, K=16, steps=5, warmup=4, cpu_baseline=True):
    """BASELINE config 2 on the GENERIC engine (engine='generic'): the per-node kernels north_star
    names -- vmp_sum_multiply / vmp_gemm_strided (Dot messages), vmp_spd_batched (GaussianARD
    moments), vmp_ewise -- driven node by node as the reference drives NumPy.  Since round 4 the
    second moments of X stay factored (Cov, <x>) instead of the reference's (1, N, K, K) array, the
    plates-sized products of the messages stay lazy, and the sweep is replayed from a HIP graph.
    This is what a model pays that misses the fused matchers."""
    import numpy as np
    import torch
    from bayespy_amd.nodes import GaussianARD, Gamma, SumMultiply
    from bayespy_amd.inference import VB
    dev = torch.device('cuda', torch.cuda.current_device())
    g = torch.Generator(device=dev)
    g.manual_seed(42)
    w = torch.randn(D, K, generator=g, device=dev, dtype=torch.float64)
    x = torch.randn(K, N, generator=g, device=dev, dtype=torch.float64)
    y = w @ x + 0.1 * torch.randn(D, N, generator=g, device=dev, dtype=torch.float64)
    x0 = torch.randn(N, K, generator=g, device=dev, dtype=torch.float64)
    del w, x
    alpha = Gamma(1e-2, 1e-2, plates=(K,), name='alpha')
    W = GaussianARD(0, alpha, shape=(K,), plates=(D, 1), name='W')
    X = GaussianARD(0, 1, shape=(K,), plates=(1, N), name='X')
    F = SumMultiply('i,i', W, X, name='F')
    tau = Gamma(1e-2, 1e-2, name='tau')
    Y = GaussianARD(F, tau, name='Y')
    X.initialize_from_value(x0[None])
    Y.observe(y)
    torch.cuda.reset_peak_memory_stats()
    Q = VB(Y, F, W, X, tau, alpha, engine='generic')
    Q.ignore_bound_checks = True
    # (two eager sweeps, then the sweep is recorded into a HIP graph and replayed --
    # plans/graph_iter.py; the warm-up covers the recording)
    Q.update(repeat=max(warmup, 4), verbose=False)
    dt, step_ms = timed_update(Q, steps)
    dt /= steps
    L = [float(v) for v in Q.L[:Q.iter]]
    # what the generic engine moves per iteration, roughly: Y read four times (two Dot messages, two
    # plate sums), <f> = W X written once and read three times, and about twenty passes over
    # (N, K) arrays (natural parameters, <x>, the terms of the bound) -- against the fused
    # block's single pass 8 N (D + K)
    alg = 8.0 * N * (D + K)
    own = 8.0 * N * (8 * D + 20 * K)
    out = {
        'metric': 'VB iterations/sec, PCA N=%d D=%d K=%d, generic engine' % (N, D, K),
        'value': 1.0 / dt, 'unit': 'VB iterations/s', 'n_gpus': 1, 'steps': steps,
        'warmup': warmup, 'ms_per_step': 1e3 * dt, 'higher_is_better': True,
        'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': 'probabilistic PCA (BASELINE config 2), N=%d D=%d K=%d, fully '
                               'observed, engine=generic (per-node kernels, no fused block)'
                               % (N, D, K), 'engine': type(Q.plans[0]).__name__,
                   'sweep_graph': Q.plans[0].graph_info()},
        'elbo_first': L[0], 'elbo_last': L[-1], 'step_ms': step_ms,
        'peak_mem_GB': torch.cuda.max_memory_allocated() / 1e9,
        'roofline': {'bound': 'hbm', 'achieved': alg / dt / 1e9, 'peak': HBM_PEAK_GBS,
                     'unit': 'GB/s', 'frac': alg / dt / 1e9 / HBM_PEAK_GBS, 'traffic': None,
                     'alg_bytes_per_iteration': alg,
                     'engine_array_bytes_per_iteration': own,
                     'engine_array_GBs': own / dt / 1e9,
                     'note': 'whole iteration (~130 launches, one graph) against the ALGORITHMIC '
                             'bytes of the fused one-pass form, 8 N (D + K); engine_array_* is an '
                             'estimate of the array passes this engine makes'},
    }
    if cpu_baseline:
        from oracle.pca import PCAOracle
        o = PCAOracle(y.cpu().numpy(), x0.cpu().numpy(), keep_x=False)
        n_it = min(3, len(L))
        t0 = time.perf_counter()
        o.iterate(n_it)
        dtc = time.perf_counter() - t0
        out['cpu_baseline'] = {
            'value': n_it / dtc, 'unit': 'VB iterations/s', 'cores': _cores(), 'kind': 'port',
            'elbo_rel_err_full': float(np.max(np.abs((np.array(o.L) - np.array(L[:n_it]))
                                                     / np.array(o.L)))),
            'sample': 'oracle/pca.py on the whole workload, %d iterations at %.2f s/iter'
                      % (n_it, dtc / n_it)}
    return out


def run_generic_gmm(N=100_000, D=16, K=32, steps=3, warmup=4, cpu_baseline=True):
    """A Gaussian mixture on the generic engine (engine='generic'; the fused block takes D <= 32,
    run_gmm(D=16) is the same model on it).  The reference forms (N, K, D, D) intermediates here
    (mixture.py:156, expfamily.py:45-61: 65 KB per point); since round 4 the engine contracts
    instead -- responsibilities phi_k . u_n as a GEMM, the messages to (mu, Lambda) as products
    summed under the mixture weights -- and replays the sweep from a HIP graph."""
    import numpy as np
    import torch
    import warnings
    from bayespy_amd.nodes import GaussianARD, Gaussian, Wishart, Dirichlet, Categorical, Mixture
    from bayespy_amd.inference import VB
    dev = torch.device('cuda', torch.cuda.current_device())
    g = torch.Generator(device=dev)
    g.manual_seed(42)
    centers = 3 * torch.randn(K, D, generator=g, device=dev, dtype=torch.float64)
    lab = torch.randint(0, K, (N,), generator=g, device=dev)
    y = centers[lab] + 0.5 * torch.randn(N, D, generator=g, device=dev, dtype=torch.float64)
    lab0 = torch.randint(0, K, (N,), generator=g, device=dev)
    alpha = Dirichlet(1e-3 * np.ones(K), name='alpha')
    z = Categorical(alpha, plates=(N,), name='z')
    mu = GaussianARD(0, 1e-3, shape=(D,), plates=(K,), name='mu')
    Lam = Wishart(D, 0.01 * np.identity(D), plates=(K,), name='Lambda')
    Y = Mixture(z, Gaussian, mu, Lam, plates=(N,), name='Y')
    z.initialize_from_value(lab0.cpu().numpy())
    Y.observe(y)
    torch.cuda.reset_peak_memory_stats()
    with warnings.catch_warnings(record=True) as wlist:
        warnings.simplefilter('always')
        Q = VB(Y, mu, Lam, z, alpha, engine='generic')
    Q.ignore_bound_checks = True
    Q.update(repeat=max(warmup, 4), verbose=False)      # covers the recording of the sweep graph
    dt, step_ms = timed_update(Q, steps)
    dt /= steps
    L = [float(v) for v in Q.L[:Q.iter]]
    FS = D * D + D + 1
    flops = 4.0 * N * K * FS
    out = {
        'metric': 'VB iterations/sec, GMM N=%d D=%d K=%d, generic engine' % (N, D, K),
        'value': 1.0 / dt, 'unit': 'VB iterations/s', 'n_gpus': 1, 'steps': steps,
        'warmup': warmup, 'ms_per_step': 1e3 * dt, 'higher_is_better': True,
        'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': 'Gaussian mixture N=%d D=%d K=%d, engine="generic": per-node kernels, '
                               'contractions instead of (N, K, D, D) intermediates' % (N, D, K),
                   'engine': type(Q.plans[0]).__name__,
                   'sweep_graph': Q.plans[0].graph_info(),
                   'matcher_said': [str(w.message)[:300] for w in wlist][:1]},
        'elbo_first': L[0], 'elbo_last': L[-1], 'step_ms': step_ms,
        'peak_mem_GB': torch.cuda.max_memory_allocated() / 1e9,
        'roofline': {'bound': 'mfma', 'achieved': flops / dt / 1e12, 'peak': FP64_MFMA_PEAK_TFLOPS,
                     'unit': 'TFLOP/s', 'frac': flops / dt / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                     'traffic': None, 'alg_flops_per_iteration': flops,
                     'note': 'whole iteration against the algorithmic flops 4 N K (D^2 + D + 1) of '
                             'SURVEY.md 8(d); this engine runs ~100 small launches per sweep '
                             '(replayed from one HIP graph), not one fused pass'},
    }
    if cpu_baseline:
        from oracle.gmm import GMMOracle
        o = GMMOracle(y.cpu().numpy(), lab0.cpu().numpy(), K)
        n_it = min(3, len(L))
        t0 = time.perf_counter()
        o.iterate(n_it, keep_r=False)
        dtc = (time.perf_counter() - t0) / n_it
        rel = max(abs(a - b) / abs(b) for a, b in zip(L[:n_it], o.L))
        out['cpu_baseline'] = {
            'value': 1.0 / dtc, 'unit': 'VB iterations/s', 'cores': _cores(), 'kind': 'port',
            'elbo_rel_err_full': float(rel),
            'sample': 'oracle/gmm.py (NumPy fp64, chunked) on the whole workload, %d iterations '
                      'at %.2f s/iter' % (n_it, dtc)}
    return out


def run_masked(N=10_000_000, D=128, K=32, steps=3, warmup=1, missing=0.1, engine=None,
               cpu_baseline=True, cpu_sample_n=200_000):
    """PCA with missing values at random (``mask = rng.rand(D, N) < 0.9``)."""
    import numpy as np
    import torch
    from bayespy_amd.nodes import GaussianARD, Gamma, SumMultiply
    from bayespy_amd.inference import VB
    dev = torch.device('cuda', torch.cuda.current_device())
    g = torch.Generator(device=dev)
    g.manual_seed(42)
    w = torch.randn(D, K, generator=g, device=dev, dtype=torch.float64)
    y = torch.empty(D, N, device=dev, dtype=torch.float64)
    mask = torch.empty(D, N, device=dev, dtype=torch.bool)
    step = 1 << 20
    for s in range(0, N, step):
        e = min(N, s + step)
        x = torch.randn(K, e - s, generator=g, device=dev, dtype=torch.float64)
        y[:, s:e] = w @ x
        y[:, s:e] += 0.1 * torch.randn(D, e - s, generator=g, device=dev, dtype=torch.float64)
        mask[:, s:e] = torch.rand(D, e - s, generator=g, device=dev) >= missing
    x0 = torch.randn(N, K, generator=g, device=dev, dtype=torch.float64)
    alpha = Gamma(1e-2, 1e-2, plates=(K,), name='alpha')
    W = GaussianARD(0, alpha, shape=(K,), plates=(D, 1), name='W')
    X = GaussianARD(0, 1, shape=(K,), plates=(1, N), name='X')
    F = SumMultiply('i,i', W, X, name='F')
    tau = Gamma(1e-2, 1e-2, name='tau')
    Y = GaussianARD(F, tau, name='Y')
    X.initialize_from_value(x0[None])
    Y.observe(y, mask=mask)
    Q = VB(Y, F, W, X, tau, alpha, engine=engine)
    Q.ignore_bound_checks = True
    plan = Q.plans[0]
    if type(plan).__name__ == 'GenericPlan' and 8.0 * N * K * K > 16e9:
        raise RuntimeError('no fused block took this model: the generic engine would '
                           'materialise (N, K, K) arrays of %.0f GB' % (8e-9 * N * K * K))
    Q.update(repeat=warmup, verbose=False)
    timed = hasattr(plan, 'enable_timing')
    if timed:
        plan.enable_timing(True)
    dt, step_ms = timed_update(Q, steps)
    # 2NDK^2 (messages to W) + 2NDK^2 (precisions of X) + NK^3/3 (Cholesky)
    flops = 4.0 * N * D * K * K + N * K ** 3 / 3.0
    L = [float(v) for v in Q.L[:Q.iter]]
    out = {
        'metric': 'VB iterations/sec, PCA N=%d D=%d K=%d, %d%% missing' % (N, D, K,
                                                                          round(100 * missing)),
        'value': steps / dt, 'unit': 'VB iterations/s', 'n_gpus': 1, 'steps': steps,
        'warmup': warmup, 'ms_per_step': 1e3 * dt / steps, 'higher_is_better': True,
        'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': 'probabilistic PCA with values missing at random (array mask), '
                               'N=%d D=%d K=%d: per-plate K x K posteriors for X and W'
                               % (N, D, K), 'engine': type(plan).__name__},
        'elbo_first': L[0], 'elbo_last': L[-1], 'step_ms': step_ms,
        'peak_mem_GB': torch.cuda.max_memory_allocated() / 1e9,
        'roofline': {'bound': 'mfma', 'achieved': None, 'peak': FP64_MFMA_PEAK_TFLOPS,
                     'unit': 'TFLOP/s', 'frac': None, 'traffic': None,
                     'achieved_alg': flops / (dt / steps) / 1e12,
                     'frac_alg': flops / (dt / steps) / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                     'alg_flops_per_iteration': flops},
    }
    # what the matrix unit EXECUTES per iteration: the two GEMM stages over the packed (symmetric)
    # columns, 2 N DP (16 (PT + KT)) flops each, and the per-plate stage, 412 instructions of 512 flops
    # per four plates at K = 32 (NB (NB + 1) (NB + 2) / 6 + ... block products; counted from the
    # kernel: SQ_INSTS_MFMA of the committed counters)
    P_ = K * (K + 1) // 2
    cols = 16 * ((P_ + 15) // 16 + (K + 15) // 16)
    DPp = 32 * max(1, 1 << max(0, (D - 1).bit_length() - 5))
    nb = (K + 3) // 4
    # per pivot block: 4 row selections + transposes and new panel blocks of the rows below + NB - 1
    # products with E + NB (NB - 1) / 2 block updates; then <x>: NB (NB + 1) / 2 + NB (412 at NB = 8)
    blk_insts = sum(4 + 2 * (nb - 1 - p) + (nb - 1) + nb * (nb - 1) // 2 for p in range(nb)) \
        + nb * (nb + 1) // 2 + nb
    issued = 2.0 * (2.0 * N * DPp * cols) + (N / 4.0) * blk_insts * 512.0
    out['roofline']['issued_mfma_flops_per_iteration'] = issued
    out['roofline']['issued_mfma_TFLOPs'] = issued / (dt / steps) / 1e12
    out['roofline']['achieved'] = out['roofline']['issued_mfma_TFLOPs']
    out['roofline']['frac'] = out['roofline']['issued_mfma_TFLOPs'] / FP64_MFMA_PEAK_TFLOPS
    out['roofline']['note'] = ('achieved / frac: flops the matrix unit EXECUTES over the whole '
                               'iteration (the GEMM stages run on the packed symmetric columns; the '
                               'per-plate stage is counted from its instruction stream, an '
                               'estimate within a few per cent); achieved_alg / frac_alg: the '
                               'ALGORITHMIC flops of SURVEY.md 8(d) (4NDK^2 + NK^3/3), an '
                               'efficiency figure')
    if timed:
        kms = plan.kernel_times_ms()
        out['roofline']['kernel_ms'] = kms
        if kms:
            ch = float(kms['chunk_plates'])
            P = K * (K + 1) // 2
            # per chunk of plates: issued (symmetry-exploiting) MFMA flops of the two GEMM stages,
            # K^3 (factor + inverse) + 2K^2 (mean) flops per plate for the sweep stage
            fused = kms['mpca_lambda'] < 0.05          # the precision GEMM inside the per-plate kernel
            gemm = 2.0 * ch * D * (P + K)
            sweep = ch * (K ** 3 + 2.0 * K * K)
            if fused:
                out['roofline']['kernels'] = [
                    {'kernel': 'mpca_blk4_kernel<fused: precision GEMM + per-plate stage>',
                     'avg_launch_ms': kms['mpca_sweep'],
                     'issued_TFLOPs': (gemm + sweep) / kms['mpca_sweep'] / 1e9,
                     'plates_per_s': ch / kms['mpca_sweep'] * 1e3}]
            else:
                out['roofline']['kernels'] = [
                    {'kernel': 'mpca_lambda_kernel', 'avg_launch_ms': kms['mpca_lambda'],
                     'issued_TFLOPs': gemm / kms['mpca_lambda'] / 1e9},
                    {'kernel': 'mpca_blk4_kernel', 'avg_launch_ms': kms['mpca_sweep'],
                     'alg_TFLOPs': sweep / kms['mpca_sweep'] / 1e9,
                     'plates_per_s': ch / kms['mpca_sweep'] * 1e3}]
            out['roofline']['kernels'].append(
                {'kernel': 'mpca_stats3_kernel+mpca_ryx_kernel', 'avg_launch_ms': kms['mpca_stats'],
                 'issued_TFLOPs': gemm / kms['mpca_stats'] / 1e9})
            out['roofline']['kernel'] = 'mpca_blk4_kernel (largest single kernel)'
    wl = 'masked PCA N=%d D=%d K=%d' % (N, D, K)
    prof, why = pmc_profile(wl)
    if prof is not None:
        out['roofline']['traffic'] = pmc_iteration_traffic(prof, 'mpca_')
        out['roofline']['traffic_source'] = prof['path'] + ' (HBM bytes per VB iteration, all ' \
                                                           'mpca_* kernels)'
    else:
        out['roofline']['traffic_source'] = why
    if cpu_baseline and type(plan).__name__ == 'MaskedPCAPlan':
        # the FIRST ns plates of the very data, mask and initial <x> of this run: the oracle
        # (timed), and the HIP path re-run on that sample from the same initial moments
        from oracle.masked_pca import MaskedPCAOracle
        ns = min(cpu_sample_n, N)
        ys = y[:, :ns].contiguous()
        ms = mask[:, :ns].contiguous()
        xs = x0[:ns].contiguous()
        del Q, plan, Y, F, W, X, tau, alpha, y, mask, x0
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        n_it = 3
        o = MaskedPCAOracle(ys.cpu().numpy(), ms.cpu().numpy(), xs.cpu().numpy(), chunk=1 << 14)
        o.iterate(1)
        t0 = time.perf_counter()
        o.iterate(n_it - 1)
        dtc = (time.perf_counter() - t0) / (n_it - 1)
        alpha = Gamma(1e-2, 1e-2, plates=(K,))
        W = GaussianARD(0, alpha, shape=(K,), plates=(D, 1))
        X = GaussianARD(0, 1, shape=(K,), plates=(1, ns))
        tau = Gamma(1e-2, 1e-2)
        Ys = GaussianARD(SumMultiply('i,i', W, X), tau)
        X.initialize_from_value(xs[None])
        Ys.observe(ys, mask=ms)
        Qs = VB(Ys, W, X, tau, alpha, engine=engine)
        Qs.ignore_bound_checks = True
        Qs.update(repeat=n_it, verbose=False)
        rel = max(abs(a - b) / abs(b) for a, b in zip(Qs.L[:n_it], o.L))
        out['cpu_baseline'] = {
            'value': 1.0 / (dtc * (N / float(ns))), 'unit': 'VB iterations/s', 'cores': _cores(),
            'kind': 'port', 'elbo_rel_err_hip_vs_oracle': float(rel),
            'elbo_iterations_compared': n_it,
            'sample': 'oracle/masked_pca.py (NumPy fp64, chunked; batched inverses on a thread '
                      'pool) on the first N=%d plates of the same data / mask / initial <x>, %d '
                      'timed iterations at %.2f s/iter, extrapolated linearly to N=%d; the '
                      'reference itself: 2.7 s/iter at N=2e4, D=64, K=16 (BASELINE.md)'
                      % (ns, n_it - 1, dtc, N)}
    return out


def run_lssm(B=100_000, T=1000, M=8, D=4, steps=3, warmup=1, cpu_baseline=True,
             cpu_sample_b=10_000):
    """Linear state-space model of bayespy/demos/lssm.py:34-103 with a sequence plate."""
    import json  # noqa: F401
    import numpy as np
    import torch
    import torch.distributed as dist
    from bayespy_amd.nodes import GaussianARD, Gamma, SumMultiply, GaussianMarkovChain
    from bayespy_amd.inference import VB
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    Bl = B * (rank + 1) // world - B * rank // world     # this rank's sequences
    dev = torch.device('cuda', torch.cuda.current_device())
    g = torch.Generator(device=dev)
    g.manual_seed(100 + rank)
    rs = np.random.RandomState(0)
    a_true = torch.from_numpy(0.9 * np.linalg.qr(rs.normal(size=(D, D)))[0]).to(dev)
    c_true = torch.from_numpy(rs.normal(size=(M, D))).to(dev)
    x = torch.empty(Bl, T, D, device=dev, dtype=torch.float64)
    x[:, 0] = torch.randn(Bl, D, generator=g, device=dev, dtype=torch.float64)
    for t in range(1, T):
        x[:, t] = x[:, t - 1] @ a_true.T + torch.randn(Bl, D, generator=g, device=dev,
                                                       dtype=torch.float64)
    y = torch.einsum('md,btd->mbt', c_true, x)
    y += 0.3 * torch.randn(M, Bl, T, generator=g, device=dev, dtype=torch.float64)
    x0 = torch.randn(Bl, T, D, generator=g, device=dev, dtype=torch.float64)
    del x
    alpha = Gamma(1e-5, 1e-5, plates=(D,), name='alpha')
    A = GaussianARD(0, alpha, shape=(D,), plates=(D,), name='A')
    A.initialize_from_value(np.identity(D))
    X = GaussianMarkovChain(np.zeros(D), 1e-3 * np.identity(D), A, np.ones(D), n=T, plates=(Bl,),
                            name='X')
    if world > 1:
        X.shard(-1)
    X.initialize_from_value(x0)
    gamma = Gamma(1e-5, 1e-5, plates=(D,), name='gamma')
    gamma.initialize_from_value(1e-2 * np.ones(D))
    C = GaussianARD(0, gamma, shape=(D,), plates=(M, 1, 1), name='C')
    c_init = rs.normal(size=(M, 1, 1, D))
    C.initialize_from_value(c_init)
    tau = Gamma(1e-5, 1e-5, name='tau')
    tau.initialize_from_value(1e2)
    F = SumMultiply('i,i', C, X, name='F')
    Y = GaussianARD(F, tau, name='Y')
    Y.observe(y)
    Q = VB(Y, F, C, gamma, X, A, alpha, tau)
    Q.ignore_bound_checks = True
    Q.update(repeat=warmup, verbose=False)
    plan = Q.plans[0]
    timed = hasattr(plan, 'kernel_times_ms')
    if timed:
        plan.enable_timing(True)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    dt, step_ms = timed_update(Q, steps, barrier)
    dt /= steps
    # algorithmic traffic per iteration: read Y (M B T), read + write the chain means (B T D)
    byts = 8.0 * B * T * (M + 2 * D)
    kms = plan.kernel_times_ms() if timed else None
    out = {
        'metric': 'VB iterations/sec, LSSM B=%d T=%d M=%d D=%d' % (B, T, M, D),
        'value': 1.0 / dt, 'unit': 'VB iterations/s', 'n_gpus': world, 'steps': steps,
        'warmup': warmup, 'ms_per_step': 1e3 * dt, 'higher_is_better': True, 'scaling': 'strong',
        'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': 'linear state-space model (GaussianMarkovChain + SumMultiply), '
                               '%d sequences x %d steps, observations %d-dim, states %d-dim'
                               % (B, T, M, D), 'engine': type(Q.plans[0]).__name__,
                   'sequences_per_rank': Bl},
        'elbo_first': float(Q.L[0]), 'elbo_last': float(Q.L[Q.iter - 1]), 'step_ms': step_ms,
        'peak_mem_GB': torch.cuda.max_memory_allocated() / 1e9,
        'roofline': {'bound': 'hbm', 'achieved': byts / dt / 1e9, 'peak': HBM_PEAK_GBS,
                     'unit': 'GB/s', 'frac': byts / dt / 1e9 / HBM_PEAK_GBS, 'traffic': None,
                     'alg_bytes_per_iteration': byts, 'kernel_ms': kms,
                     'note': 'whole iteration against the ALGORITHMIC bytes: read Y once + '
                             'read/write <x> once'},
    }
    prof, why = pmc_profile('LSSM B=%d T=%d M=%d D=%d' % (B, T, M, D))
    if prof is not None and world == 1:
        out['roofline']['traffic'] = pmc_iteration_traffic(prof, 'lssm_')
        out['roofline']['traffic_source'] = prof['path'] + ' (HBM bytes per VB iteration, all ' \
                                                           'lssm_* kernels)'
    else:
        out['roofline']['traffic_source'] = why
    if cpu_baseline and world == 1 and type(plan).__name__ == 'LSSMPlan':
        # the FIRST bs sequences of the very data and initial moments of this run: the oracle
        # (timed) and the HIP path re-run on that sample
        from oracle.lssm import LSSMOracle
        bs = min(cpu_sample_b, B)
        ys = y[:, :bs].contiguous()
        xs = x0[:bs].contiguous()
        del Q, plan, Y, F, C, gamma, X, A, alpha, tau, y, x0
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        n_it = 3
        o = LSSMOracle(ys.cpu().numpy(), xs.cpu().numpy(), c_init.reshape(M, D))
        o.iterate(1)
        t0 = time.perf_counter()
        o.iterate(n_it - 1)
        dtc = (time.perf_counter() - t0) / (n_it - 1)
        alpha = Gamma(1e-5, 1e-5, plates=(D,))
        A = GaussianARD(0, alpha, shape=(D,), plates=(D,))
        A.initialize_from_value(np.identity(D))
        X = GaussianMarkovChain(np.zeros(D), 1e-3 * np.identity(D), A, np.ones(D), n=T,
                                plates=(bs,))
        X.initialize_from_value(xs)
        gamma = Gamma(1e-5, 1e-5, plates=(D,))
        gamma.initialize_from_value(1e-2 * np.ones(D))
        C = GaussianARD(0, gamma, shape=(D,), plates=(M, 1, 1))
        C.initialize_from_value(c_init)
        tau = Gamma(1e-5, 1e-5)
        tau.initialize_from_value(1e2)
        Ys = GaussianARD(SumMultiply('i,i', C, X), tau)
        Ys.observe(ys)
        Qs = VB(Ys, C, gamma, X, A, alpha, tau)
        Qs.ignore_bound_checks = True
        Qs.update(repeat=n_it, verbose=False)
        rel = max(abs(a - b) / abs(b) for a, b in zip(Qs.L[:n_it], o.L))
        out['cpu_baseline'] = {
            'value': 1.0 / (dtc * (B / float(bs))), 'unit': 'VB iterations/s', 'cores': 1,
            'kind': 'port', 'elbo_rel_err_hip_vs_oracle': float(rel),
            'elbo_iterations_compared': n_it,
            'sample': 'oracle/lssm.py (NumPy fp64: one shared covariance recursion + vectorised '
                      'mean recursions over the sequences, single-threaded apart from BLAS) on '
                      'the first B=%d sequences of the same data / initial moments, %d timed '
                      'iterations at %.2f s/iter, extrapolated linearly to B=%d; the reference '
                      'itself: ~3.6e2 s/iter extrapolated (BASELINE.md section 2)'
                      % (bs, n_it - 1, dtc, B)}
    return out


def build_lssm_masked(B=10_000, T=1000, M=8, D=4, missing=0.3, seed=7):
    """The state-space model of run_lssm observed through a per-sequence mask (30 % missing at
    random plus a stretch of 50 steps without any data, as bayespy/demos/lssm.py:239-246 does on
    its single chain); returns (VB, dict with the data / mask / initial values)."""
    import numpy as np
    import torch
    from bayespy_amd.nodes import GaussianARD, Gamma, SumMultiply, GaussianMarkovChain
    from bayespy_amd.inference import VB
    dev = torch.device('cuda', torch.cuda.current_device())
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    rs = np.random.RandomState(0)
    a_true = torch.from_numpy(0.9 * np.linalg.qr(rs.normal(size=(D, D)))[0]).to(dev)
    c_true = torch.from_numpy(rs.normal(size=(M, D))).to(dev)
    x = torch.empty(B, T, D, device=dev, dtype=torch.float64)
    x[:, 0] = torch.randn(B, D, generator=g, device=dev, dtype=torch.float64)
    for t in range(1, T):
        x[:, t] = x[:, t - 1] @ a_true.T + torch.randn(B, D, generator=g, device=dev,
                                                       dtype=torch.float64)
    y = torch.einsum('md,btd->mbt', c_true, x)
    y += 0.3 * torch.randn(M, B, T, generator=g, device=dev, dtype=torch.float64)
    mask = torch.rand(M, B, T, generator=g, device=dev) >= missing
    lo = min(30, max(T - 2, 0))
    mask[:, :, lo:min(lo + 50, T - 1)] = False
    x0 = torch.randn(B, T, D, generator=g, device=dev, dtype=torch.float64)
    del x
    alpha = Gamma(1e-5, 1e-5, plates=(D,), name='alpha')
    A = GaussianARD(0, alpha, shape=(D,), plates=(D,), name='A')
    A.initialize_from_value(np.identity(D))
    X = GaussianMarkovChain(np.zeros(D), 1e-3 * np.identity(D), A, np.ones(D), n=T, plates=(B,),
                            name='X')
    X.initialize_from_value(x0)
    gamma = Gamma(1e-5, 1e-5, plates=(D,), name='gamma')
    gamma.initialize_from_value(1e-2 * np.ones(D))
    C = GaussianARD(0, gamma, shape=(D,), plates=(M, 1, 1), name='C')
    c_init = rs.normal(size=(M, 1, 1, D))
    C.initialize_from_value(c_init)
    tau = Gamma(1e-5, 1e-5, name='tau')
    tau.initialize_from_value(1e2)
    F = SumMultiply('i,i', C, X, name='F')
    Y = GaussianARD(F, tau, name='Y')
    Y.observe(y, mask=mask)
    Q = VB(Y, F, C, gamma, X, A, alpha, tau)
    Q.ignore_bound_checks = True
    return Q, dict(y=y, mask=mask, x0=x0, c_init=c_init)


def run_lssm_masked(B=10_000, T=1000, M=8, D=4, steps=50, warmup=2, cpu_baseline=True,
                    cpu_sample_b=64):
    """SURVEY.md 8(f).2 with array masks: B sequences x T steps, one mask per sequence."""
    import numpy as np
    import torch
    Q, info = build_lssm_masked(B, T, M, D)
    plan = Q.plans[0]
    Q.update(repeat=warmup, verbose=False)
    timed = type(plan).__name__ == 'MaskedLSSMPlan'
    if timed:
        plan.enable_timing(True)
    dt, step_ms = timed_update(Q, steps)
    dt /= steps
    NS = D * (D + 1) // 2
    # algorithmic traffic per iteration: read Y and the mask word once, write <x> and <x x^T> once
    # (per-sequence second moments are part of the answer here: every sequence has its own)
    byts = 8.0 * B * T * (M + 1 + D + NS)
    kms = plan.kernel_times_ms() if timed else None
    out = {
        'metric': 'VB iterations/sec, masked LSSM B=%d T=%d M=%d D=%d' % (B, T, M, D),
        'value': 1.0 / dt, 'unit': 'VB iterations/s', 'n_gpus': 1, 'steps': steps,
        'warmup': warmup, 'ms_per_step': 1e3 * dt, 'higher_is_better': True, 'scaling': 'strong',
        'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': 'linear state-space model observed through one mask per sequence '
                               '(30 %% missing + a stretch without data), %d sequences x %d steps, '
                               'observations %d-dim, states %d-dim' % (B, T, M, D),
                   'engine': type(plan).__name__},
        'elbo_first': float(Q.L[0]), 'elbo_last': float(Q.L[Q.iter - 1]), 'step_ms': step_ms,
        'peak_mem_GB': torch.cuda.max_memory_allocated() / 1e9,
        'roofline': {'bound': 'hbm', 'achieved': byts / dt / 1e9, 'peak': HBM_PEAK_GBS,
                     'unit': 'GB/s', 'frac': byts / dt / 1e9 / HBM_PEAK_GBS, 'traffic': None,
                     'alg_bytes_per_iteration': byts, 'kernel_ms': kms,
                     # forward: Y, mask in, F out; backward (statistics carried): F, Y, mask in, <x>, <xx> out
                     'moved_bytes_per_iteration': 8.0 * B * T * (2 * (M + 1) + 3 * (NS + D))},
    }
    prof, why = pmc_profile('masked LSSM B=%d T=%d M=%d D=%d' % (B, T, M, D))
    if prof is not None:
        out['roofline']['traffic'] = pmc_iteration_traffic(prof, 'lssmm_')
        out['roofline']['traffic_source'] = prof['path']
    else:
        out['roofline']['traffic_source'] = why
    if cpu_baseline and timed:
        # the FIRST bs sequences of the very data, mask and initial moments: oracle (timed) and the
        # HIP path re-run on that sample
        from oracle.lssm import MaskedLSSMOracle
        from bayespy_amd.nodes import GaussianARD, Gamma, SumMultiply, GaussianMarkovChain
        from bayespy_amd.inference import VB
        bs = min(cpu_sample_b, B)
        ys = info['y'][:, :bs].contiguous()
        ms = info['mask'][:, :bs].contiguous()
        xs = info['x0'][:bs].contiguous()
        c_init = info['c_init']
        del Q, plan, info
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        n_it = 3
        o = MaskedLSSMOracle(ys.cpu().numpy(), ms.cpu().numpy(), xs.cpu().numpy(),
                             c_init.reshape(M, D))
        o.iterate(1)
        t0 = time.perf_counter()
        o.iterate(n_it - 1)
        dtc = (time.perf_counter() - t0) / (n_it - 1)
        alpha = Gamma(1e-5, 1e-5, plates=(D,))
        A = GaussianARD(0, alpha, shape=(D,), plates=(D,))
        A.initialize_from_value(np.identity(D))
        X = GaussianMarkovChain(np.zeros(D), 1e-3 * np.identity(D), A, np.ones(D), n=T,
                                plates=(bs,))
        X.initialize_from_value(xs)
        gamma = Gamma(1e-5, 1e-5, plates=(D,))
        gamma.initialize_from_value(1e-2 * np.ones(D))
        C = GaussianARD(0, gamma, shape=(D,), plates=(M, 1, 1))
        C.initialize_from_value(c_init)
        tau = Gamma(1e-5, 1e-5)
        tau.initialize_from_value(1e2)
        Ys = GaussianARD(SumMultiply('i,i', C, X), tau)
        Ys.observe(ys, mask=ms)
        Qs = VB(Ys, C, gamma, X, A, alpha, tau)
        Qs.ignore_bound_checks = True
        Qs.update(repeat=n_it, verbose=False)
        rel = max(abs(a - b) / abs(b) for a, b in zip(Qs.L[:n_it], o.L))
        out['cpu_baseline'] = {
            'value': 1.0 / (dtc * (B / float(bs))), 'unit': 'VB iterations/s', 'cores': 1,
            'kind': 'port', 'elbo_rel_err_hip_vs_oracle': float(rel),
            'elbo_iterations_compared': n_it,
            'sample': 'oracle/lssm.py:MaskedLSSMOracle (NumPy fp64, batched over the sequences) on '
                      'the first B=%d sequences of the same data / mask / initial moments, %d '
                      'timed iterations at %.2f s/iter, extrapolated linearly to B=%d'
                      % (bs, n_it - 1, dtc, B)}
    return out
