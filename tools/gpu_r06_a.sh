#!/bin/bash
# round 6, step A: the fused shared-covariance update + contraction planning of the generic engine
O=gpurun_out/r06_a
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_generic_gpu.py -x -q -k "gaussian_shared_update" > $O/pytest_kernel.log 2>&1
tail -5 $O/pytest_kernel.log
timeout 1500 python -m pytest tests/test_generic_engine_gpu.py tests/test_generic_gpu.py tests/test_graph_sweep_gpu.py tests/test_update_order_gpu.py tests/test_hyperparameters_gpu.py tests/test_input_forms_gpu.py tests/test_sharded_generic_gpu.py -q > $O/pytest_generic.log 2>&1
tail -15 $O/pytest_generic.log
timeout 300 python bench.py --config generic_pca --no-cpu-baseline > $O/bench_generic_pca.json 2> $O/bench_generic_pca.err
timeout 300 python bench.py --config generic_gmm --no-cpu-baseline > $O/bench_generic_gmm.json 2> $O/bench_generic_gmm.err
tail -c 600 $O/bench_generic_pca.json; tail -3 $O/bench_generic_pca.err
tail -c 600 $O/bench_generic_gmm.json; tail -3 $O/bench_generic_gmm.err
