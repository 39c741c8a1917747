#!/bin/bash
# Usage (on the GPU box, from the repo root): scripts/profile_round.sh <tag>
# Everything profiles/ holds for a round, from ONE box and the tree as it is:
#   gpurun_out/<tag>_bench.json         the bench line (no profiler)
#   gpurun_out/prof_<tag>/              rocprofv3 --kernel-trace --stats of the same command
#   gpurun_out/pmc_<tag>/ + _summary    the PMC passes (scripts/pmc_passes.sh, separate runs)
#   gpurun_out/<tag>_traffic.json       HBM bytes per launch and step (scripts/make_traffic.py)
set -u
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
python3 "$ROOT/bench.py" --steps 5 --warmup 2 > "$OUT/${TAG}_bench.json" 2> "$OUT/${TAG}_bench.err" || tail -3 "$OUT/${TAG}_bench.err"
echo "bench done"
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_$TAG" -- \
   python3 "$ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-shortcut-leg > "$OUT/prof_$TAG.json" 2> "$OUT/prof_$TAG.err") \
   || tail -3 "$OUT/prof_$TAG.err"
echo "kernel trace done"
bash "$ROOT/scripts/pmc_passes.sh" "$TAG"
python3 "$ROOT/scripts/pmc_summary.py" "$OUT/pmc_$TAG" > "$OUT/pmc_${TAG}_summary.txt"
python3 "$ROOT/scripts/make_traffic.py" "$OUT/pmc_$TAG" "$OUT/prof_$TAG" "$OUT/prof_$TAG.json" > "$OUT/${TAG}_traffic.json"
cp "$OUT"/prof_$TAG/*/*kernel_stats.csv "$OUT/${TAG}_kernel_stats.csv" 2>/dev/null
echo "profile set $TAG done"
