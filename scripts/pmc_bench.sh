#!/bin/bash
# PMC passes over bench.py for profiles/<R>_igemm_traffic.json and <R>_mfma_utilisation.json.  Counters are collected in their
# own runs (no trace domains besides the kernel trace), one counter set per pass (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE
# do not fit one pass).  Serialized kernels (--no-overlap) so that a dispatch's counters are its own.
export TMPDIR=/tmp
R=${1:-r02}
for MATH in split f32; do
  for CNT in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    tag=${MATH}_$(echo $CNT | cut -d' ' -f1)
    d=gpurun_out/pmc_$R/$tag; mkdir -p $d
    timeout -k 10 280 rocprofv3 --kernel-trace --pmc $CNT --output-format csv -d $d -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt --no-unfused --no-overlap --math $MATH > $d/stdout.json 2> $d/stderr.log || { tail -5 $d/stderr.log; exit 1; }
    echo "$tag done"
  done
done
python3 scripts/pmc_summarize.py $R
