#!/bin/bash
# rocprofv3 kernel-trace summaries of bench.py for profiles/: f32 overlapped, f32 serialized, split overlapped.
export TMPDIR=/tmp
R=${1:-r01f}
for V in "f32_overlap:--math f32" "f32_serialized:--math f32 --no-overlap" "split_overlap:--math split" "split_serialized:--math split --no-overlap"; do
  tag=${V%%:*}; args=${V#*:}
  d=gpurun_out/prof_$R/$tag; mkdir -p $d
  timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o p -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-alt --no-unfused $args > $d/stdout.json 2> $d/stderr.log || exit 1
  echo "$tag done"
done
