#!/usr/bin/env python3
"""Does the constant-line shortcut fire?  One axis pass over a 512^3 volume of +0, -0, 1 and
noise, IFE_OPT_CONST_LINES on and off; prints the pass's kernel time."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
pkg = importlib.import_module("image-feature-extraction_amd")
ctx = pkg.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
shape = (512, 512, 512)
d_out = torch.empty(shape, dtype=torch.float32, device="cuda")
vols = {"+0": torch.zeros(shape, device="cuda"), "-0": -torch.zeros(shape, device="cuda"),
        "1": torch.ones(shape, device="cuda"), "noise": torch.randn(shape, device="cuda")}
names = {0: "iir_x", 1: "iir_y", 2: "iir_z"}
for axis in (2, 0, 1):
    for name, v in vols.items():
        row = []
        for opt in (0, 1):
            ctx.set_option(pkg.OPT_CONST_LINES, opt)
            for _ in range(2):
                ctx.stage_recursive_gaussian(v.data_ptr(), d_out.data_ptr(), shape, (1, 1, 1), axis, 2.0)
            ctx.set_option(pkg.OPT_PROFILE, 1)
            ctx.reset_kernel_times()
            for _ in range(5):
                ctx.stage_recursive_gaussian(v.data_ptr(), d_out.data_ptr(), shape, (1, 1, 1), axis, 2.0)
            ctx.synchronize()
            n, ms = ctx.kernel_times()[names[axis]]
            ctx.set_option(pkg.OPT_PROFILE, 0)
            row.append(ms / n)
        print("axis %d %-5s off %.3f ms  on %.3f ms" % (axis, name, row[0], row[1]), flush=True)

# the pipeline's shape: six jobs of one launch, three on noise and three on a constant field
outs = [torch.empty(shape, dtype=torch.float32, device="cuda") for _ in range(6)]
for cname in ("1", "+0"):
    ins = [vols["noise"], vols[cname]] * 3
    for axis in (2, 0):
        row = []
        for opt in (0, 1):
            ctx.set_option(pkg.OPT_CONST_LINES, opt)
            call = lambda: ctx.stage_recursive_gaussian_batch([t.data_ptr() for t in ins], [t.data_ptr() for t in outs],
                                                              shape, (1, 1, 1), axis, [1.0, 1.0, 2.0, 2.0, 4.0, 4.0])
            call(); call()
            ctx.set_option(pkg.OPT_PROFILE, 1)
            ctx.reset_kernel_times()
            for _ in range(3):
                call()
            ctx.synchronize()
            n, ms = ctx.kernel_times()[names[axis]]
            ctx.set_option(pkg.OPT_PROFILE, 0)
            row.append(ms / n)
        print("six jobs, axis %d, noise + %-2s: off %.3f ms  on %.3f ms" % (axis, cname, row[0], row[1]), flush=True)

# the whole path on a CT-like volume: every value outside the mask negative (air), so that the
# numerator's exterior is -0 and the denominator's +0; bench.py's ellipsoid mask, 3 scales
del outs
synth = importlib.import_module("image-feature-extraction_amd.synthetic")
mask = np.minimum(synth.mask_ellipsoids(shape), 1).astype(np.uint8)
img = synth.volume_f32(shape, synth.SEED_CONFIG[3])
img = np.where(mask != 0, img, -1000.0 - np.abs(img) * 0.01).astype(np.float32)
d_img, d_mask = torch.from_numpy(img).cuda(), torch.from_numpy(mask).cuda()
d_feat = torch.empty((3,) + shape + (8,), dtype=torch.float32, device="cuda")
import time
for opt in (0, 1, 0, 1):
    ctx.set_option(pkg.OPT_CONST_LINES, opt)
    run = lambda: ctx.emphysema_features_device(d_img.data_ptr(), pkg.F32, d_mask.data_ptr(), pkg.U8, shape,
                                                (1, 1, 1), [1.0, 2.0, 4.0], d_feat.data_ptr())
    run(); run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    print("CT-like exterior, whole path, const_lines %d: %.3f ms per step" % (opt, (time.perf_counter() - t0) / 5 * 1e3), flush=True)
