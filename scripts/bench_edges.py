#!/usr/bin/env python3
"""Row f1 measurement: one image of DetermineHistogramBinEdges_MultiScaleEigenvalueFeatures
(all-foreground branch) on one MI355X: a5 at every scale -> gather of the foreground
samples -> radix sort of the 8*S columns -> equalizing edges.  Prints one JSON line.

  python scripts/bench_edges.py [--size NZ NY NX] [--sigmas ...] [--bins 41] [--steps 5]

Not the driver's bench (that is bench.py, the a1-a9 path); same conventions: inputs
resident in HBM, hipEvent time per kernel kind from inside the library, the oracle timed
on the host cores on a bounded sample."""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "image-feature-extraction_amd"
HBM_PEAK_GBS = 8000.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, nargs=3, default=[512, 512, 512])
    ap.add_argument("--sigmas", type=float, nargs="+", default=[1.0, 2.0, 4.0])
    ap.add_argument("--bins", type=int, default=41)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--mask", choices=["ellipsoids", "ones"], default="ellipsoids")
    ap.add_argument("--cpu-sample", type=int, default=160, help="edge of the CPU baseline cube")
    args = ap.parse_args()
    import torch
    pkg = importlib.import_module(PKG)
    synth = importlib.import_module(PKG + ".synthetic")
    shape = tuple(args.size)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    img = synth.volume_f32(shape, synth.SEED_CONFIG[3])
    lab = synth.mask_ellipsoids(shape) if args.mask == "ellipsoids" else np.ones(shape, np.uint8)
    fg = (1, 2) if args.mask == "ellipsoids" else (1,)
    d_img, d_lab = torch.from_numpy(img).to(dev), torch.from_numpy(lab).to(dev)
    ctx = pkg.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    s = ctx.samples(8 * len(args.sigmas))

    def step():
        s.clear()
        s.add_image_device(d_img.data_ptr(), pkg.F32, d_lab.data_ptr(), pkg.U8, shape, args.sigmas, fg)
        return s.equalized_edges(args.bins)

    for _ in range(args.warmup):
        step()
    ctx.set_option(pkg.OPT_PROFILE, 1)
    ctx.reset_kernel_times()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        edges = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    kt = ctx.kernel_times()
    ctx.set_option(pkg.OPT_PROFILE, 0)
    nvox = int(np.prod(shape))
    nsamp = s.count(0)
    ncol = 8 * len(args.sigmas)
    kern = {k: {"launches_per_step": n / args.steps, "ms_per_step": round(ms / args.steps, 4)}
            for k, (n, ms) in kt.items()}
    sort_ms = sum(kern[k]["ms_per_step"] for k in ("sort_hist", "sort_scan", "sort_scatter") if k in kern)
    keys = nsamp * ncol
    moved = keys * 12 * 4          # per pass: read (histogram) + read + write (scatter), 4 passes
    out = {
        "metric": "Mvoxels/sec features -> equalized histogram edges (row f1)",
        "value": round(nvox * len(args.sigmas) / dt / 1e6, 1), "unit": "Mvoxels/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt * 1e3, 3),
        "higher_is_better": True, "dtype": "u32 keys (order-preserving map of f32)", "data": "synthetic",
        "config": {"workload": "%dx%dx%d float32, sigma=%s, labels %s foreground %s (%.1f%% of voxels), "
                               "%d columns x %d samples, %d bins"
                               % (shape[2], shape[1], shape[0], args.sigmas, args.mask, list(fg),
                                  100.0 * nsamp / nvox, ncol, nsamp, args.bins)},
        "roofline": {"bound": "hbm", "scope": "radix sort (4 passes x 3 kernels) of all columns",
                     "achieved": round(keys * 8 / (sort_ms * 1e-3) / 1e9, 1) if sort_ms else None,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(keys * 8 / (sort_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if sort_ms else None,
                     "algorithmic_bytes": keys * 8, "moved_bytes_by_design": moved,
                     "moved_GBs": round(moved / (sort_ms * 1e-3) / 1e9, 1) if sort_ms else None,
                     "sort_Gkeys_per_s": round(keys / (sort_ms * 1e-3) / 1e9, 2) if sort_ms else None,
                     "kernels": kern},
    }
    # CPU baseline: the oracle doing the same on a cube of the same volume
    from oracle import pyoracle as O
    O.build()
    th = min(os.cpu_count() or 1, 16)
    O.set_threads(th)
    e = args.cpu_sample
    ci, cl = img[:e, :e, :e].copy(), lab[:e, :e, :e].copy()
    t1 = time.perf_counter()
    for sg in args.sigmas:
        f = O.emphysema_features(ci, np.minimum(cl, 1).astype(np.uint8), sg)
        g = O.gather_foreground(f, cl, fg)
        for c in range(8):
            if g.shape[1] >= args.bins:
                O.equalized_edges(O.sort_f32(g[c]), args.bins)
    cdt = time.perf_counter() - t1
    out["cpu_baseline"] = {"value": round(e ** 3 * len(args.sigmas) / cdt / 1e6, 3), "unit": "Mvoxels/s",
                           "cores": th, "kind": "port",
                           "sample": "%d^3 corner, %.1f s (features on %d threads; gather, qsort and "
                                     "edges scalar as in the reference tool)" % (e, cdt, th)}
    out["edges_checksum"] = float(np.asarray(edges, np.float64).sum())
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
