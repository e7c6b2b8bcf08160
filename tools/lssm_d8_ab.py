"""A/B of the state-space block, same data, same process:
  * 7 / 8 states: plate sums in the backward sweep's registers (lssm_split_from = 9: those instances
    spill) against the split form (state-only register sweeps + matrix-core sums; = 7, the default);
  * 9 .. 16 states: the workgroup form of the matrix-core sums (lssm_stats_form = 0) against the
    one-wavefront form that reads the smoothed states once (= 1, the default)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools import workloads
from bayespy_amd.device import get_runtime
rt = get_runtime()
def run(D, B, M=8):
    r = workloads.run_lssm(B=B, T=1000, M=M, D=D, steps=8, warmup=2, cpu_baseline=False)
    torch.cuda.empty_cache()
    return r['ms_per_step']
for D in (8, 7):
    for B in (20000, 100000):
        for frm, form in ((9, 1), (7, 0), (7, 1)):
            rt.lib.vmp_tune_set(b'lssm_split_from', frm)
            rt.lib.vmp_tune_set(b'lssm_stats_form', form)
            print('D=%d B=%d  split_from=%d stats_form=%d: %.3f ms per iteration' % (D, B, frm, form, run(D, B)), flush=True)
rt.lib.vmp_tune_set(b'lssm_split_from', 7)
for D, M in ((16, 8), (12, 8), (16, 32)):
    for B in (20000, 100000):
        if M == 32 and B == 100000:
            continue
        for form in (0, 1):
            rt.lib.vmp_tune_set(b'lssm_stats_form', form)
            print('D=%d M=%d B=%d  stats_form=%d: %.3f ms per iteration' % (D, M, B, form, run(D, B, M)), flush=True)
