#!/bin/bash
# Usage (on the GPU box, from the repo root): scripts/pmc_passes.sh <tag> [bench args...]
# Runs bench.py under rocprofv3 once per counter group (separate --pmc passes, kernel
# trace only) and leaves CSVs under gpurun_out/pmc_<tag>/<group>/.
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
declare -A CGRP
CGRP[sq1]="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES"
CGRP[sq2]="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS"
CGRP[tcc1]="TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"
CGRP[fetch]="FETCH_SIZE"
CGRP[write]="WRITE_SIZE GRBM_GUI_ACTIVE"
for g in sq1 sq2 tcc1 fetch write; do
  rocprofv3 --kernel-trace --pmc ${CGRP[$g]} --output-format csv -d "$OUT/$g" -- \
    python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-shortcut-leg "$@" > "$OUT/$g.json" 2> "$OUT/$g.err" \
    || { echo "pass $g failed"; tail -5 "$OUT/$g.err"; }
  echo "pass $g done"
done
