#!/bin/bash
# PMC counter passes over a small python driver (default scripts/bench_stem.py): one counter set per pass, kernel trace only.
# usage: scripts/pmc_kernels.sh <tag> <script.py> [args...]; summary -> gpurun_out/pmc_<tag>/summary.txt
export TMPDIR=/tmp
TAG=${1:-stem}; shift
SCRIPT=${1:-scripts/bench_stem.py}; shift
OUT=gpurun_out/pmc_$TAG; mkdir -p $OUT
SETS=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
      "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
      "FETCH_SIZE" "WRITE_SIZE")
i=0
for CNT in "${SETS[@]}"; do
  d=$OUT/set$i; mkdir -p $d
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $CNT --output-format csv -d $d -o p -- python3 $SCRIPT "$@" > $d/stdout.txt 2> $d/stderr.log || { echo "set $i failed"; tail -3 $d/stderr.log; }
  i=$((i+1))
done
python3 scripts/pmc_kernels_summarize.py $OUT > $OUT/summary.txt
cat $OUT/summary.txt
