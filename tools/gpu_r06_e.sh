#!/bin/bash
O=gpurun_out/r06_e
mkdir -p $O
export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_pca_gpu.py tests/test_comm_gpu.py tests/test_doubles_pinned_gpu.py tests/test_empty_plates_gpu.py -q -x > $O/pytest.log 2>&1
tail -4 $O/pytest.log
for i in 1 2 3; do timeout 300 python bench.py --config pca_c2 --steps 200 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2 fused', d['ms_per_step'])"; done
for i in 1 2; do VMP_PCA_FUSE_GRAM=0 timeout 300 python bench.py --config pca_c2 --steps 200 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2 unfused', d['ms_per_step'])"; done
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline', d['ms_per_step'], d['value'], d['roofline']['frac'])"
timeout 300 python bench.py --n 1250000 --steps 100 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('shard 1.25e6', d['ms_per_step'])"
VMP_PCA_FUSE_GRAM=0 timeout 300 python bench.py --n 1250000 --steps 100 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('shard 1.25e6 unfused', d['ms_per_step'])"
