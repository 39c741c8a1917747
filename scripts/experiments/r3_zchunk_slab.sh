#!/bin/bash
show() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().splitlines()[-1]); print('$1', d['ms_per_step'], {k:round(v['ms_per_step'],3) for k,v in d['roofline']['kernels'].items()})"; }
for zc in 64 32 22 16 11; do
python3 bench.py --no-cpu-baseline --force-slab --size 64 512 512 --line-groups 2 --zchunk $zc 2>/dev/null | show "64 planes zchunk=$zc"
done
for zc in 64 43 32; do
python3 bench.py --no-cpu-baseline --force-slab --size 128 512 512 --line-groups 4 --zchunk $zc 2>/dev/null | show "128 planes zchunk=$zc"
done
