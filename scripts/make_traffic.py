#!/usr/bin/env python3
"""HBM bytes per launch and per step from the PMC passes of scripts/pmc_passes.sh.

Usage: scripts/make_traffic.py gpurun_out/pmc_<tag> gpurun_out/prof_<tag> bench.json > profiles/rNN_traffic.json

read  = 2 x FETCH_SIZE KiB (gfx950 counts 128-B requests as 64 B: MI355X_MICROARCH.md, HBM section)
write = WRITE_SIZE KiB
launches_per_step comes from the kernel-trace stats of the same bench command
(calls / (steps + warmup)); the algorithmic bytes are the bench line's own figure.
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

pmc_root, prof_root, bench_json = sys.argv[1:4]


def short(name):
    name = re.sub(r"\(.*", "", name)
    return re.sub(r"^void ", "", name)


ctr = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(pmc_root, "*", "*", "*counter_collection.csv")):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if row["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
                ctr[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))

bench = json.loads(open(bench_json).read().strip().splitlines()[-1])
passes = bench["steps"] + bench["warmup"]
kernels = {}
total = 0.0
for f in glob.glob(os.path.join(prof_root, "*", "*kernel_stats.csv")):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = short(row["Name"])
            if k not in ctr or not k.startswith("ife::"):
                continue  # the bench's own copy-bandwidth probe is not part of the step
            fetch = ctr[k]["FETCH_SIZE"]
            write = ctr[k]["WRITE_SIZE"]
            rd = 2.0 * 1024.0 * sum(fetch) / len(fetch)
            wr = 1024.0 * sum(write) / len(write)
            per_step = int(row["Calls"]) / passes
            kernels[k] = {"avg_ms": float(row["AverageNs"]) * 1e-6, "read_bytes": rd,
                          "write_bytes": wr, "launches_per_step": per_step}
            total += (rd + wr) * per_step

print(json.dumps({
    "note": "per-launch HBM traffic from rocprofv3 PMC (separate passes): read = 2 x FETCH_SIZE KiB "
            "(gfx950 correction), write = WRITE_SIZE KiB; workload: " + bench["config"]["workload"],
    "kernels": kernels,
    "traffic_bytes_per_step": total,
    "algorithmic_bytes_per_step": bench["roofline"].get("algorithmic_bytes_per_step", 14898167808),
}, indent=1))
