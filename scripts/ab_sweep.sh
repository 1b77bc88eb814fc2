#!/bin/bash
# Same-box A/B: scripts/ab_sweep.sh OUTDIR NAME1 NAME2 ...  -> two interleaved rounds of scripts/layer_sweep.py per build_ab/NAME.so
OUT=$1; shift
mkdir -p $OUT
for r in 1 2; do
  for n in "$@"; do
    MLA_HIP_LIB=$PWD/build_ab/$n.so timeout -k 10 200 python scripts/layer_sweep.py > $OUT/${n}_$r.txt 2>&1 || exit 1
    echo "$n round $r: $(tail -n 1 $OUT/${n}_$r.txt)"
  done
done
