"""A/B of the state-space block at 7 / 8 states: the register-resident sweeps (plate sums in the
sweep's registers: the D = 8 instance spills) against the matrix-core path of 8 < D <= 16 (tune key
lssm_mfma_from = 7), same data, same process; parity of both against the oracle at a small size."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tools import workloads
from bayespy_amd.device import get_runtime
rt = get_runtime()
for D in (8, 7):
    for B in (20000, 100000):
        for frm in (9, 7):
            rt.lib.vmp_tune_set(b'lssm_mfma_from', frm)
            r = workloads.run_lssm(B=B, T=1000, M=8, D=D, steps=8, warmup=2, cpu_baseline=False)
            print('D=%d B=%d  %s: %.3f ms per iteration' % (D, B, 'matrix-core path' if frm == 7 else 'register sweeps  ', r['ms_per_step']))
            torch.cuda.empty_cache()
rt.lib.vmp_tune_set(b'lssm_mfma_from', 9)
