#!/bin/bash
# rocprofv3 evidence of round 2 (summaries only travel back).  Usage: tools/collect_profiles_r02.sh <outdir>
O=${1:-gpurun_out/prof_r02}
mkdir -p $O
R=$PWD
export TMPDIR=/tmp
BID=$(python -c "import sys; sys.path.insert(0,'$R'); from bayespy_amd import _lib; print(_lib.load().vmp_version().decode().split('build ')[-1])")
prof() {  # name, only-filter, command...
  local name=$1 only=$2; shift 2
  (cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p_st_$name -o r -- "$@" > $R/$O/under_rocprof_$name.log 2>&1)
  python tools/rocpd_summary.py /tmp/p_st_$name/r_results.db > $O/kernel_stats_$name.txt 2>&1
}
pmcs() {  # name, only-filter, workload-string, command...
  local name=$1 only=$2 wl=$3; shift 3
  local dbs=""
  local i=0
  for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU" "GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
    i=$((i+1))
    (cd /tmp; timeout 600 rocprofv3 --pmc $set --kernel-trace -d /tmp/p_pmc_${name}_$i -o r -- "$@" > /dev/null 2>&1)
    dbs="$dbs /tmp/p_pmc_${name}_$i/r_results.db"
  done
  ( echo "# build_id: $BID"; echo "# workload: $wl"; python tools/rocpd_summary.py --pmc --only $only $dbs ) > $O/pmc_$name.txt 2>&1
}
B="python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra"
prof pca_gram pass_kernel $B
pmcs pca_gram pass_kernel "D=128 K=32 n_local=10000000" $B
M="python $R/tools/bench_masked_pca.py --n 2000000 --steps 2"
prof masked mpca $M
pmcs masked mpca "masked PCA N=2000000 D=128 K=32" $M
Ls="python $R/bench.py --config lssm --steps 3"
prof lssm lssm $Ls
pmcs lssm lssm "LSSM B=100000 T=1000 M=8 D=4" $Ls
G="python $R/bench.py --config gmm --steps 5 --no-cpu-baseline"
prof gmm gmm_pass $G
pmcs gmm gmm_pass "GMM N=10000000 D=8 K=64" $G
ls -la $O
