#!/bin/bash
# Collect the rocprofv3 evidence for profiles/ on the GPU box (summaries only; the
# rocpd databases are too large to travel back).  Usage: tools/collect_profiles.sh <outdir>
O=${1:-gpurun_out/prof}
mkdir -p $O
R=$PWD
export TMPDIR=/tmp
for st in gram stream; do
  B="python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --stats $st"
  (cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_stats_$st -o r -- $B > $R/$O/bench_under_rocprof_$st.log 2>&1)
  python tools/rocpd_summary.py /tmp/p_stats_$st/r_results.db > $O/kernel_stats_pca_$st.txt 2>&1
  (cd /tmp; timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/p_fetch_$st -o r -- $B > /dev/null 2>&1)
  (cd /tmp; timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/p_write_$st -o r -- $B > /dev/null 2>&1)
  (cd /tmp; timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d /tmp/p_sq_$st -o r -- $B > /dev/null 2>&1)
  (cd /tmp; timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS --kernel-trace -d /tmp/p_sq2_$st -o r -- $B > /dev/null 2>&1)
  python tools/rocpd_summary.py --pmc /tmp/p_fetch_$st/r_results.db /tmp/p_write_$st/r_results.db /tmp/p_sq_$st/r_results.db /tmp/p_sq2_$st/r_results.db > $O/pmc_pca_$st.txt 2>&1
done
G="python $R/tools/bench_gmm.py --steps 5 --no-cpu-baseline"
(cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_stats_gmm -o r -- $G > $R/$O/bench_under_rocprof_gmm.log 2>&1)
python tools/rocpd_summary.py /tmp/p_stats_gmm/r_results.db > $O/kernel_stats_gmm.txt 2>&1
(cd /tmp; timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --kernel-trace -d /tmp/p_sq_gmm -o r -- $G > /dev/null 2>&1)
(cd /tmp; timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/p_f_gmm -o r -- $G > /dev/null 2>&1)
(cd /tmp; timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/p_w_gmm -o r -- $G > /dev/null 2>&1)
python tools/rocpd_summary.py --pmc /tmp/p_sq_gmm/r_results.db /tmp/p_f_gmm/r_results.db /tmp/p_w_gmm/r_results.db > $O/pmc_gmm.txt 2>&1
# masked PCA (generic engine) and the shard-size step (what one of 8 ranks runs at N=1e7)
M="python $R/tools/bench_masked_pca.py --n 200000 --steps 9"
(cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_stats_mpca -o r -- $M > $R/$O/bench_under_rocprof_masked_pca.log 2>&1)
python tools/rocpd_summary.py /tmp/p_stats_mpca/r_results.db > $O/kernel_stats_masked_pca.txt 2>&1
python tools/bench_masked_pca.py --n 20000 --steps 10 > $O/bench_masked_pca_n2e4.json 2>/dev/null
python tools/bench_masked_pca.py --n 200000 --steps 10 > $O/bench_masked_pca_n2e5.json 2>/dev/null
S="python $R/bench.py --n 1250000 --steps 10 --warmup 2 --no-cpu-baseline"
(cd /tmp; timeout 300 rocprofv3 --kernel-trace -d /tmp/p_tl -o r -- $S > /dev/null 2>&1)
python tools/rocpd_summary.py --timeline 30 /tmp/p_tl/r_results.db > $O/timeline_shard_n1250000.txt 2>&1
python bench.py --n 1250000 --no-cpu-baseline > $O/bench_shard_n1250000_overlap.json 2>/dev/null
VMP_PCA_PLATE_STREAM=0 python bench.py --n 1250000 --no-cpu-baseline > $O/bench_shard_n1250000_inorder.json 2>/dev/null
VMP_PCA_PLATE_STREAM=0 python bench.py --no-cpu-baseline > $O/bench_n1_gram_inorder.json 2>/dev/null
python bench.py --n 1000000 --d 64 --k 16 --no-cpu-baseline > $O/bench_config2.json 2>/dev/null
python tools/bench_lssm.py > $O/bench_lssm.json 2>/dev/null
python bench.py --steps 20 --warmup 3 > $O/bench_n1_gram.json 2>/dev/null
python bench.py --steps 20 --warmup 3 --stats stream --no-cpu-baseline > $O/bench_n1_stream.json 2>/dev/null
python tools/bench_gmm.py > $O/bench_gmm_n1.json 2>/dev/null
tools/microbench.bin > $O/microbench.txt 2>&1
# forward-backward kernel of categorical Markov chains and the generic kernels (incl. the
# matrix-core SPD sweep)
H="python $R/tools/bench_hmm.py --reps 3"
(cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_stats_hmm -o r -- $H > $R/$O/bench_under_rocprof_hmm.log 2>&1)
python tools/rocpd_summary.py /tmp/p_stats_hmm/r_results.db > $O/kernel_stats_hmm.txt 2>&1
(python tools/bench_hmm.py; python tools/bench_hmm.py --k 3 --chains 50000; python tools/bench_hmm.py --k 16 --chains 8000 --steps 500; python tools/bench_hmm.py --chains 1 --steps 100000) > $O/bench_hmm.json 2>/dev/null
python tools/bench_generic.py > $O/bench_generic.json 2>/dev/null
python tools/bench_masked_pca.py --n 200000 --d 128 --k 32 --steps 5 > $O/bench_masked_pca_n2e5_d128_k32.json 2>/dev/null
ls -la $O
