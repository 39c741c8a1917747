#!/bin/bash
# round 3: what the per-kernel hipEvents of bench.py's timed region cost (two records per launch)
show() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().splitlines()[-1]); print('$1', d['ms_per_step'])"; }
run() { python3 bench.py --no-cpu-baseline --no-shortcut-leg --no-stream-probe --steps 20 --warmup 3 "$@" 2>gpurun_out/ev.err; }
for rep in 1 2 3; do
  run --proxy-world 8 --proxy-rank 3 | show "rank 3 of 8, events"
  run --proxy-world 8 --proxy-rank 3 --no-kernel-events | show "rank 3 of 8, no events"
  run | show "single device, events"
  run --no-kernel-events | show "single device, no events"
done
