#!/bin/bash
# round 3: feature kernel (ring form) at reduced occupancy: extra dynamic LDS per workgroup
# (static 30.7 KB: 0 -> 3 workgroups per CU by the launch bound, 33 KB -> 2, 60 KB -> 1)
show() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().splitlines()[-1]); print('$1', d['ms_per_step'], {k:v['avg_ms'] for k,v in d['roofline']['kernels'].items()})"; }
for rep in 1 2; do
  for x in 0 33000 60000; do
    IFE_DIAG_FT_DYNLDS=$x python3 bench.py --no-cpu-baseline --no-shortcut-leg --no-stream-probe 2>gpurun_out/occ.err | show "dynlds=$x"
  done
done
