#!/bin/bash
# round 3: A/B of the current library against saved builds under scripts/experiments/libs (one box, one call)
set -u
show() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().splitlines()[-1]); print('$1', d['ms_per_step'], {k:v['avg_ms'] for k,v in d['roofline']['kernels'].items()})"; }
shopt -s nullglob
for rep in 1 2; do
  python3 bench.py --no-cpu-baseline --no-shortcut-leg "$@" 2>gpurun_out/ab.err | show "current"
  for v in scripts/experiments/libs/*.so; do
    IFE_HIP_LIB=$v python3 bench.py --no-cpu-baseline --no-shortcut-leg "$@" 2>gpurun_out/ab.err | show "$(basename $v)"
  done
done
