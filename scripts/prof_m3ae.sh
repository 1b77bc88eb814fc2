#!/bin/bash
# rocprofv3 kernel-trace of the M3AE (config 4) MLA step.  OVERLAP=0 serialises the encoder chains so that per-kernel
# durations are not inflated by co-running kernels; TAG names the output.
export TMPDIR=/tmp; TAG=${TAG:-m3}; mkdir -p gpurun_out/prof_m3ae
MATH=${MATH:-f32} STEPS=3 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_m3ae -o $TAG -- python3 scripts/bench_m3ae.py > gpurun_out/prof_m3ae/${TAG}_stdout.log 2> gpurun_out/prof_m3ae/${TAG}_stderr.log
tail -1 gpurun_out/prof_m3ae/${TAG}_stdout.log
