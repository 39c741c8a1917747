#!/usr/bin/env python3
"""Summarise the CSVs left by scripts/pmc_passes.sh: per kernel name, mean counter value
per dispatch.  Usage: scripts/pmc_summary.py gpurun_out/pmc_<tag>"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "*", "*", "*counter_collection.csv")):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = re.sub(r"\(.*", "", row["Kernel_Name"])
            k = re.sub(r"^void ", "", k)
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
names = sorted({c for k in acc for c in acc[k]})
for k in sorted(acc):
    print(k)
    for c in names:
        if c in acc[k]:
            v = acc[k][c]
            print("   %-24s mean %.4g  (n=%d)" % (c, sum(v) / len(v), len(v)))
