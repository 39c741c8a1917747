#!/bin/bash
# bench.py over the default library and every variant under scripts/experiments/libs/
show() { python -c "
import json,sys
d=json.loads(sys.stdin.read().splitlines()[-1]); print('$1', d['ms_per_step'], {k:v['avg_ms'] for k,v in d['roofline']['kernels'].items()})"; }
python bench.py --no-cpu-baseline "$@" 2>/dev/null | show default
for v in scripts/experiments/libs/*.so; do IFE_HIP_LIB=$v python bench.py --no-cpu-baseline "$@" 2>/dev/null | show $(basename $v); done
