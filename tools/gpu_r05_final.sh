#!/bin/bash
# Round-5 closing evidence on one box: the GPU suite, the default bench line (+ verbose records), a world-1 RCCL
# launch, the kernel traces of replayed generic sweeps with and without the queue of small operations.
O=gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tail -6 > $O/pytest_gpu_summary.txt
python bench.py --full-out $O/bench_full.json > $O/bench_default.json 2> $O/bench_default.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 1 --steps 5 --warmup 2 --no-extra --no-cpu-baseline --steady-steps 0 > $O/bench_world1_nccl.json 2> $O/bench_world1.err
GRAPH=1 BAYESPY_AMD_GRAPH_QUEUE=0 NLAST=1500 tools/queue_lab_prof.sh $PWD/$O/queue_prof_a > $O/queue_lab_prof_noqueue.txt 2>&1
GRAPH=1 BAYESPY_AMD_GRAPH_QUEUE=1 NLAST=1500 tools/queue_lab_prof.sh $PWD/$O/queue_prof_b > $O/queue_lab_prof_queue.txt 2>&1
tail -3 $O/pytest_gpu_summary.txt; tail -c 600 $O/bench_default.json; tail -3 $O/queue_lab_prof_noqueue.txt
