#!/usr/bin/env python3
"""HBM bytes per launch and per step from the PMC passes of scripts/pmc_passes.sh.

Usage: scripts/make_traffic.py gpurun_out/pmc_<tag> gpurun_out/prof_<tag> bench.json > profiles/rNN_traffic.json

read  = 2 x FETCH_SIZE KiB (gfx950 counts 128-B requests as 64 B: MI355X_MICROARCH.md, HBM section)
write = WRITE_SIZE KiB
"kinds" repeats the figures under bench.py's kernel-kind names (iir_z, iir_x, iir_y, features,
prep) together with SQ_INSTS_VALU per launch, which bench.py turns into the issue roofline.
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
            if row["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU", "GRBM_GUI_ACTIVE"):
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
            valu = ctr[k].get("SQ_INSTS_VALU")
            if valu:
                kernels[k]["valu_insts"] = sum(valu) / len(valu)
            gui = ctr[k].get("GRBM_GUI_ACTIVE")
            if gui:  # summed over the 8 XCDs; per second of the (profiled) launch
                kernels[k]["clock_GHz"] = sum(gui) / len(gui) / 8.0 / float(row["AverageNs"])
            total += (rd + wr) * per_step

def kind_of(name):
    """bench.py's kernel-kind name of a profiled kernel of the default workload."""
    if "features" in name:
        return "features"
    if "iir_contig" in name:
        return "iir_x"
    if "iir_strided_kernel" in name:
        return "iir_y" if ", true>" in name else "iir_z"
    if "prep_kernel" in name:
        return "prep"
    return None


kinds = {}
for k, v in kernels.items():
    kd = kind_of(k)
    if kd and kd not in kinds:
        kinds[kd] = dict(v, kernel=k)

print(json.dumps({
    "note": "per-launch HBM traffic from rocprofv3 PMC (separate passes): read = 2 x FETCH_SIZE KiB "
            "(gfx950 correction), write = WRITE_SIZE KiB; workload: " + bench["config"]["workload"],
    "kernels": kernels,
    "kinds": kinds,
    "traffic_bytes_per_step": total,
    "algorithmic_bytes_per_step": bench["roofline"].get("algorithmic_bytes_per_step", 14898167808),
}, indent=1))
