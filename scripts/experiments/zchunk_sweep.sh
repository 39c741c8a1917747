#!/bin/bash
# feature-kernel z-chunk sweep: 64-plane slab (the per-rank volume at 8 GPUs) and full size
show() { python -c "
import json,sys
d=json.loads(sys.stdin.read().splitlines()[-1]); print('$1', d['ms_per_step'], d['roofline']['kernels']['features']['avg_ms'])"; }
for zc in 64 32 16 8; do python bench.py --no-cpu-baseline --size 64 512 512 --zchunk $zc 2>/dev/null | show "slab64 zchunk=$zc"; done
for zc in 64 32 16; do python bench.py --no-cpu-baseline --zchunk $zc 2>/dev/null | show "full zchunk=$zc"; done
