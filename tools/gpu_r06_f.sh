#!/bin/bash
run() { echo -n "$1: "; env $1 timeout 300 python bench.py --config pca_c2 --steps 300 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
run VMP_PCA_FUSE_GRAM=0
run VMP_PCA_FUSE_GRAM=1
run "VMP_PCA_FUSE_GRAM=0 VMP_PCA_XPASS_WGS_PER_CU=2"
run "VMP_PCA_FUSE_GRAM=1 VMP_PCA_XPASS_WGS_PER_CU=2"
run "VMP_PCA_FUSE_GRAM=0 VMP_PCA_XPASS_WGS_PER_CU=1"
run "VMP_PCA_FUSE_GRAM=1 VMP_PCA_XPASS_WGS_PER_CU=1"
run "VMP_PCA_FUSE_GRAM=0 VMP_PCA_RESERVE_CUS=8"
run "VMP_PCA_FUSE_GRAM=1 VMP_PCA_RESERVE_CUS=8"
run "VMP_PCA_FUSE_GRAM=0 VMP_PCA_PLATE_STREAM=0"
