#!/bin/bash
# round 3: per-rank timing proxies with neighbours (bench.py --proxy-world: slab.NullComm, no transfers):
# interior rank 3 of 8 and edge rank 0 of 8, for the current library and every build under libs/
show() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().splitlines()[-1]); print('$1', d['ms_per_step'], {k:round(v['ms_per_step'],3) for k,v in d['roofline']['kernels'].items()})"; }
run() { python3 bench.py --no-cpu-baseline --steps 20 --warmup 3 "$@" 2>gpurun_out/proxy.err; }
shopt -s nullglob
for rep in 1 2; do
  for lib in current scripts/experiments/libs/*.so; do
    [ "$lib" = current ] && unset IFE_HIP_LIB || export IFE_HIP_LIB=$lib
    for r in 3 0; do
      run --proxy-world 8 --proxy-rank $r | show "$(basename $lib) rank $r of 8"
    done
    run --force-slab --size 64 512 512 | show "$(basename $lib) 64 planes, no neighbour"
  done
done
unset IFE_HIP_LIB
run --proxy-world 4 --proxy-rank 1 | show "current rank 1 of 4"
run --proxy-world 2 --proxy-rank 0 | show "current rank 0 of 2"
