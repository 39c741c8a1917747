#!/usr/bin/env python3
"""Instruction histogram of one kernel in csrc/ife_capi.gfx950.s (`make -C csrc asm`).

usage: asm_hist.py <regex on the mangled name> [--loops]
Prints, per matching kernel: register use, and the number of VALU (split f64 / other),
SALU, LDS and vector-memory instructions in the whole body and, with --loops, per basic
block that ends in a backward branch (the loop bodies).  Static counts: a tuning aid,
not a measurement.
"""
import collections
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ASM = os.path.join(HERE, "..", "image-feature-extraction_amd", "csrc", "ife_capi.gfx950.s")


def classify(op):
    if op.startswith("v_"):
        return "valu_f64" if "f64" in op else "valu"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "vmem"
    return None


def main():
    pat = re.compile(sys.argv[1])
    loops = "--loops" in sys.argv
    text = open(ASM).read()
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)^\s*s_endpgm", text, re.S | re.M):
        name, body = m.group(1), m.group(2)
        if not pat.search(name):
            continue
        tot = collections.Counter()
        blocks, cur, label = [], collections.Counter(), "entry"
        labels_seen = {}
        for i, line in enumerate(body.split("\n")):
            t = line.strip()
            lm = re.match(r"^(\.LBB\w+):", t)
            if lm:
                blocks.append((label, cur))
                label, cur = lm.group(1), collections.Counter()
                labels_seen[label] = len(blocks)
                continue
            om = re.match(r"^([a-z_0-9]+)", t)
            if not om:
                continue
            k = classify(om.group(1))
            if k:
                tot[k] += 1
                cur[k] += 1
            bm = re.match(r"^s_cbranch\w*\s+(\.LBB\w+)", t) or re.match(r"^s_branch\s+(\.LBB\w+)", t)
            if bm and bm.group(1) in labels_seen and loops:
                # backward branch: everything from that label to here is one loop body
                start = labels_seen[bm.group(1)]
                acc = collections.Counter()
                for _, c in blocks[start:]:
                    acc.update(c)
                acc.update(cur)
                print("    loop %-14s %s" % (bm.group(1), dict(acc)))
        blocks.append((label, cur))
        meta = re.search(r"\.amdhsa_kernel %s\n(.*?)\.end_amdhsa_kernel" % re.escape(name), text, re.S)
        vg = re.search(r"\.amdhsa_next_free_vgpr (\d+)", meta.group(1)).group(1) if meta else "?"
        lds = re.search(r"\.amdhsa_group_segment_fixed_size (\d+)", meta.group(1)).group(1) if meta else "?"
        print("%s\n  vgpr %s lds %s  %s" % (name, vg, lds, dict(tot)))


if __name__ == "__main__":
    main()
