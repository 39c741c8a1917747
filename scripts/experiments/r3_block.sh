show() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().splitlines()[-1]); print('$1', d['ms_per_step'], {k:v['avg_ms'] for k,v in d['roofline']['kernels'].items()})"; }
for rep in 1 2 3; do for b in 0 12 16; do python3 bench.py --no-cpu-baseline --no-shortcut-leg --no-stream-probe --iir-block $b 2>/dev/null | show "block=$b"; done; done
