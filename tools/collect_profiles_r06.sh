#!/bin/bash
# rocprofv3 evidence of round 6 (summaries only travel back).  Usage: tools/collect_profiles_r04.sh <outdir> [names...]
# kernel-trace stats and, in SEPARATE runs, the PMC passes (FETCH_SIZE / WRITE_SIZE / SQ_*), every
# file headed by the build id of the library, the workload string bench.py matches, and the number
# of VB iterations the profiled command ran.  Every step runs under its own `timeout`.
O=${1:-gpurun_out/prof_r06}
shift
WHICH=${@:-pca_gram gmm masked lssm lssm_d8 lssm_d16 lssm_masked lssm_masked_1e5 generic_pca generic_gmm}
mkdir -p $O
R=$PWD
export TMPDIR=/tmp
BID=$(timeout 120 python -c "import sys; sys.path.insert(0,'$R'); from bayespy_amd import _lib; print(_lib.load().vmp_version().decode().split('build ')[-1])")
prof() {  # name, steady-state kernel substring, its launches in the iterations, command...
  local name=$1 sk=$2 sn=$3; shift 3
  (cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_st_$name -o r -- "$@" > $R/$O/under_rocprof_$name.log 2>&1)
  ( echo "# build_id: $BID"; timeout 120 python tools/rocpd_summary.py --steady $sk $sn /tmp/p_st_$name/r_results.db ) > $O/kernel_stats_$name.txt 2>&1
}
pmcs() {  # name, only-filter, workload-string, iterations, command...
  local name=$1 only=$2 wl=$3 its=$4; shift 4
  local dbs=""
  local i=0
  for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU" "GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
    i=$((i+1))
    (cd /tmp; timeout 300 rocprofv3 --pmc $set --kernel-trace -d /tmp/p_pmc_${name}_$i -o r -- "$@" > /dev/null 2>&1)
    dbs="$dbs /tmp/p_pmc_${name}_$i/r_results.db"
  done
  ( echo "# build_id: $BID"; echo "# workload: $wl"; echo "# iterations: $its"; timeout 120 python tools/rocpd_summary.py --pmc --only $only $dbs ) > $O/pmc_$name.txt 2>&1
}
for w in $WHICH; do
  case $w in
    pca_gram)
      B="python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra --steady-steps 0"
      prof pca_gram pca_xpass_kernel 12 $B
      pmcs pca_gram pass_kernel "D=128 K=32 n_local=10000000" 12 $B ;;
    masked)
      M="python $R/bench.py --config masked --exact-steps --steps 5 --warmup 1 --no-cpu-baseline"
      prof masked mpca_blk4 60 $M
      pmcs masked mpca_ "masked PCA N=10000000 D=128 K=32" 6 $M ;;
    lssm)
      Ls="python $R/bench.py --config lssm --exact-steps --steps 5 --warmup 1 --no-cpu-baseline"
      Ll="python $R/bench.py --config lssm --exact-steps --steps 50 --warmup 1 --no-cpu-baseline"
      prof lssm lssm_backward_ck 50 $Ll
      pmcs lssm lssm_ "LSSM B=100000 T=1000 M=8 D=4" 6 $Ls ;;
    lssm_d8)
      L8="python $R/bench.py --config lssm_d8 --exact-steps --steps 5 --warmup 1 --no-cpu-baseline"
      L8l="python $R/bench.py --config lssm_d8 --exact-steps --steps 30 --warmup 1 --no-cpu-baseline"
      prof lssm_d8 lssm_stats_wave 90 $L8l
      pmcs lssm_d8 lssm_ "LSSM B=100000 T=1000 M=8 D=8" 6 $L8 ;;
    lssm_d16)
      L16="python $R/bench.py --config lssm_d16 --exact-steps --steps 5 --warmup 1 --no-cpu-baseline"
      L16l="python $R/bench.py --config lssm_d16 --exact-steps --steps 30 --warmup 1 --no-cpu-baseline"
      prof lssm_d16 lssm_backward_mfma 30 $L16l
      pmcs lssm_d16 lssm_ "LSSM B=100000 T=1000 M=8 D=16" 6 $L16 ;;
    lssm_masked)
      Lm="python $R/bench.py --config lssm_masked --exact-steps --steps 5 --warmup 1 --no-cpu-baseline"
      Lml="python $R/bench.py --config lssm_masked --exact-steps --steps 50 --warmup 1 --no-cpu-baseline"
      prof lssm_masked lssmm_backward 50 $Lml
      pmcs lssm_masked lssmm_ "masked LSSM B=10000 T=1000 M=8 D=4" 6 $Lm ;;
    lssm_masked_1e5)
      Lb="python $R/bench.py --config lssm_masked_1e5 --exact-steps --steps 4 --warmup 1 --no-cpu-baseline"
      prof lssm_masked_1e5 lssmm_backward 5 $Lb
      pmcs lssm_masked_1e5 lssmm_ "masked LSSM B=100000 T=1000 M=8 D=4" 5 $Lb ;;
    generic_pca)
      # the generic engine's sweep replayed from its HIP graph: per-kernel times of the ~130 launches
      Gp="python $R/bench.py --config generic_pca --exact-steps --steps 50 --no-cpu-baseline"
      prof generic_pca gshared_pass_kernel 50 $Gp
      pmcs generic_pca gshared_pass "PCA N=1000000 D=64 K=16 generic engine" 54 $Gp ;;
    generic_gmm)
      Gg="python $R/bench.py --config generic_gmm --exact-steps --steps 50 --no-cpu-baseline"
      prof generic_gmm gemm_kernel 100 $Gg ;;
    gmm)
      G="python $R/bench.py --config gmm --exact-steps --steps 50 --warmup 2 --no-cpu-baseline"
      Gs="python $R/bench.py --config gmm --exact-steps --steps 5 --warmup 2 --no-cpu-baseline"
      prof gmm gmm_pass_kernel 50 $G
      pmcs gmm gmm_pass "GMM N=10000000 D=8 K=64" 7 $Gs ;;
  esac
done
ls -la $O
