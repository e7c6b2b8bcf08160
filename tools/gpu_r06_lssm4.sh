#!/bin/bash
python - <<'PY'
import sys, gc
sys.path.insert(0,'.')
import torch
from tools import workloads
for (B,D,M) in ((98304,16,8),(100000,16,8),(65536,16,8),(66000,16,8),(131072,16,8)):
    r = workloads.run_lssm(B=B, T=1000, M=M, D=D, steps=8, warmup=2, cpu_baseline=False)
    print('B=%d D=%d M=%d: %.3f ms per iteration = %.2f ns per sequence-step' % (B, D, M, r['ms_per_step'], r['ms_per_step']*1e6/(B*1000)), flush=True)
    del r; gc.collect(); torch.cuda.empty_cache()
PY
