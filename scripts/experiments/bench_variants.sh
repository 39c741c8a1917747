#!/bin/bash
# usage: scratch/bench_variants.sh lib1 lib2 ... ; prints features/iir kernel avg ms per variant
for lib in "$@"; do
  if [ "$lib" = "default" ]; then unset IFE_HIP_LIB; else export IFE_HIP_LIB=$GRAFT_REPO_ROOT/scripts/experiments/libs/$lib; fi
  for zc in 32 64 128; do
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --zchunk $zc 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']
print('$lib zchunk=$zc', 'step', d['ms_per_step'], ' '.join('%s=%.3f'%(n,k[n]['avg_ms']) for n in k))"
  done
done
