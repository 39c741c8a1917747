#!/bin/bash
# round 3: ring vs staged feature kernel, z-chunk sweep (one box, one call)
set -u
OUT=gpurun_out
show() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().splitlines()[-1]); print('$1', d['ms_per_step'], {k:v['avg_ms'] for k,v in d['roofline']['kernels'].items()})"; }
for r in 1 0 1 0; do
  python3 bench.py --no-cpu-baseline --no-shortcut-leg --feat-ring $r 2>$OUT/ab.err | show "ring=$r"
done
for zc in 32 43 64 86 128 171; do
  python3 bench.py --no-cpu-baseline --no-shortcut-leg --feat-ring 1 --zchunk $zc 2>$OUT/ab.err | show "ring=1 zchunk=$zc"
done
for zc in 86 128; do
  python3 bench.py --no-cpu-baseline --no-shortcut-leg --feat-ring 0 --zchunk $zc 2>$OUT/ab.err | show "ring=0 zchunk=$zc"
done
