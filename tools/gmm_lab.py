"""Lab of the mixture pass: phase 2 on the long (v_mfma_f64_16x16x4) against the short
(v_mfma_f64_4x4x4, four blocks) matrix instruction, same box, same data (tune key "gmm_mfma4").

    python tools/gmm_lab.py [--N 10000000] [--D 8] [--K 64] [--steps 30]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--N', type=int, default=10_000_000)
    ap.add_argument('--D', type=int, default=8)
    ap.add_argument('--K', type=int, default=64)
    ap.add_argument('--steps', type=int, default=30)
    args = ap.parse_args()
    from bayespy_amd import _lib
    from tools import workloads
    lib = _lib.load()
    for m4 in (0, 1, 0, 1):
        lib.vmp_tune_set(b'gmm_mfma4', m4)
        r = workloads.run_gmm(N=args.N, D=args.D, K=args.K, steps=args.steps, warmup=3,
                              cpu_baseline=False)
        roof = r['roofline']
        print(json.dumps({'gmm_mfma4': m4, 'N': args.N, 'D': args.D, 'K': args.K,
                          'ms_per_iter': r['ms_per_step'], 'pass_ms': roof.get('avg_launch_ms'),
                          'frac_alg': roof.get('frac_alg'), 'elbo_last': r.get('elbo_last')}),
              flush=True)


if __name__ == '__main__':
    main()
