#!/bin/bash
O=gpurun_out/r06_lssm
mkdir -p $O
timeout 1800 python -m pytest tests/test_lssm_gpu.py tests/test_chain_gpu.py -q -x > $O/pytest.log 2>&1
tail -15 $O/pytest.log
