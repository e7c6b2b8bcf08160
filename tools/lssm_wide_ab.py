"""A/B of the state-space block at 7 .. 16 states, every variant in ONE process on the same data, by
tune key (ms per VB iteration, T = 1000, M = 8):
  lssm_split_from   9: D = 7 / 8 carry the plate sums in the backward sweep's registers (those instances
                       spill); 7 (default): state-only sweeps + the sums as a matrix-core pass
  lssm_stats_form   0: workgroup form of that pass (two barriers per tile, states read twice);
                    1 (default): one wavefront per (32 sequences, time chunk)
  lssm_fuse_project 0: projected data h = tau C^T y as an array; 1 (default): inside the forward sweep
  lssm_segments     0: covariance recursion, then the sweeps; 1 (default): in time segments side by side
  lssm_big_mfma     0: 8 < D sweeps as one thread per sequence; 1 (default): on the matrix cores
  lssm_fuse_stats   0: plate sums of 8 < D as a pass of their own; 1: inside the backward sweep; 2 (default):
                       inside it for D >= 15 and B >= 4e4
  lssm_sweep_waves  4: four wavefronts (128 sequences) per workgroup of the matrix-core sweeps; 1 (default): one
  lssm_cov_mfma     0: 8 < D covariance recursion by 256 threads through LDS; 1 (default): one wavefront,
                       block sweeps on the matrix cores"""
import gc, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools import workloads
from bayespy_amd.device import get_runtime
rt = get_runtime()
DEFAULTS = {b'lssm_split_from': 7, b'lssm_stats_form': 1, b'lssm_fuse_project': 1, b'lssm_segments': 1,
            b'lssm_big_mfma': 1, b'lssm_cov_mfma': 1, b'lssm_fuse_stats': 2, b'lssm_sweep_waves': 1}


def run(D, B, M=8, **keys):
    for k, v in DEFAULTS.items():
        rt.lib.vmp_tune_set(k, keys.get(k.decode(), v))
    r = workloads.run_lssm(B=B, T=1000, M=M, D=D, steps=8, warmup=2, cpu_baseline=False)
    ms = r['ms_per_step']
    del r
    gc.collect()
    torch.cuda.empty_cache()
    return ms


print('library', rt.lib.vmp_version().decode())
for D in (8, 7):
    for B in (20000, 100000):
        for keys in (dict(lssm_split_from=9), dict(lssm_stats_form=0), dict()):
            print('D=%2d B=%6d  %-24s %7.3f ms' % (D, B, keys or 'defaults', run(D, B, **keys)), flush=True)
for D in (16, 12):
    for B in (20000, 100000):
        for keys in (dict(lssm_big_mfma=0), dict(lssm_stats_form=0), dict(lssm_fuse_project=0),
                     dict(lssm_segments=0), dict(lssm_cov_mfma=0), dict(lssm_fuse_stats=0), dict(lssm_fuse_stats=1), dict(lssm_sweep_waves=4), dict()):
            print('D=%2d B=%6d  %-24s %7.3f ms' % (D, B, keys or 'defaults', run(D, B, **keys)), flush=True)
for k, v in DEFAULTS.items():
    rt.lib.vmp_tune_set(k, v)
