#!/bin/bash
# PMC pass over the M3AE (config 4) step: MFMA-busy per kernel.  Counters in their own run, program after `--`.
export TMPDIR=/tmp
for MATH in split f32; do
  d=gpurun_out/pmc_m3ae/$MATH; mkdir -p $d
  OVERLAP=0 MATH=$MATH STEPS=2 timeout -k 10 280 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $d -o p -- python3 scripts/bench_m3ae.py > $d/stdout.log 2> $d/stderr.log || { tail -5 $d/stderr.log; exit 1; }
  echo "$MATH done: $(tail -1 $d/stdout.log)"
done
