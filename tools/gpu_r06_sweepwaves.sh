#!/bin/bash
timeout 900 python -m pytest tests/test_lssm_gpu.py tests/test_chain_gpu.py -x -q -m gpu 2>&1 | tail -2
python - <<'PY'
import sys, gc
sys.path.insert(0,'.')
import torch
from tools import workloads
from bayespy_amd.device import get_runtime
rt = get_runtime()
for (B,D,M) in ((100000,16,8),(66000,16,8),(65536,16,8),(20000,16,8),(100000,12,8),(100000,9,8)):
    for sw in (4, 1):
        rt.lib.vmp_tune_set(b'lssm_sweep_waves', sw)
        r = workloads.run_lssm(B=B, T=1000, M=M, D=D, steps=8, warmup=2, cpu_baseline=False)
        print('B=%d D=%d M=%d sweep_waves=%d: %.3f ms per iteration' % (B, D, M, sw, r['ms_per_step']), flush=True)
        del r; gc.collect(); torch.cuda.empty_cache()
PY
