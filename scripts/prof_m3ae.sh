#!/bin/bash
# rocprofv3 kernel-trace of the M3AE (config 4) MLA step
export TMPDIR=/tmp; mkdir -p gpurun_out/prof_m3ae
MATH=${MATH:-f32} STEPS=3 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_m3ae -o m3 -- python3 scripts/bench_m3ae.py > gpurun_out/prof_m3ae/stdout.log 2> gpurun_out/prof_m3ae/stderr.log
tail -1 gpurun_out/prof_m3ae/stdout.log
