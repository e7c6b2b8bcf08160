#!/bin/bash
timeout 900 python -m pytest tests/test_lssm_gpu.py tests/test_chain_gpu.py -x -q -m gpu 2>&1 | tail -12
