#!/bin/bash
# per-rank proxies of the slab engine on one GPU (no neighbour): wall ms per step
show() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().splitlines()[-1]); print('$1', d['ms_per_step'], {k:round(v['ms_per_step'],3) for k,v in d['roofline']['kernels'].items()})"; }
python3 bench.py --no-cpu-baseline --force-slab --size 64 512 512 --line-groups 4 2>/dev/null | show "64 planes (W=8), 4 groups, all scales/item"
python3 bench.py --no-cpu-baseline --force-slab --size 64 512 512 --line-groups 2 2>/dev/null | show "64 planes, 2 groups, all scales/item"
python3 bench.py --no-cpu-baseline --force-slab --size 64 512 512 --line-groups 3 --scales-per-item 1 2>/dev/null | show "64 planes, 3 groups, 1 scale/item (the default of round 2)"
python3 bench.py --no-cpu-baseline --force-slab --size 128 512 512 --line-groups 4 2>/dev/null | show "128 planes (W=4), 4 groups"
python3 bench.py --no-cpu-baseline --force-slab --size 256 512 512 --line-groups 4 2>/dev/null | show "256 planes (W=2), 4 groups"
python3 bench.py --no-cpu-baseline --no-shortcut-leg --no-stream-probe 2>/dev/null | show "single device, whole volume"
