"""BASELINE configs[4] at full size on one GPU: non-finite values per feature column and scale,
and for the first scale the error against the oracle per block of planes (diagnostic for
tests/test_gpu_fullsize.py::test_config5_full_size_matches_oracle)."""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
ife = importlib.import_module("image-feature-extraction_amd")
synth = importlib.import_module("image-feature-extraction_amd.synthetic")
from oracle import pyoracle as oracle  # noqa: E402  (checker)

oracle.build()
shape = (768, 1024, 1024)
spacing = (0.7, 0.7, 1.0)
sigmas = [1.0, 2.0, 3.0, 4.0, 6.0]
img = synth.volume_i16(shape, synth.SEED_CONFIG[5])
mask = np.empty(shape, np.uint8)
for z in range(0, shape[0], 64):
    mask[z:z + 64] = np.minimum(synth.mask_ellipsoids((min(64, shape[0] - z),) + shape[1:], z0=z,
                                                      nz_total=shape[0]), 1)
print("foreground", float(mask.mean()), flush=True)
oracle.set_threads(min(16, os.cpu_count() or 1))
with ife.Context(0) as c:
    for s, got in enumerate(c.emphysema_features_stream(img, mask, sigmas, spacing)):
        bad = [int((~np.isfinite(got[..., k])).sum()) for k in range(8)]
        print("sigma", sigmas[s], "non-finite per column", bad, flush=True)
        if s == 0:
            ref = oracle.emphysema_features(img.astype(np.float32), mask, sigmas[s], spacing)
            for z in range(0, shape[0], 64):
                g = got[z:z + 64].reshape(-1, 8).astype(np.float64)
                r = ref[z:z + 64].reshape(-1, 8).astype(np.float64)
                lam = np.maximum(np.abs(r[:, 2]), 1e-30)
                e = np.abs(np.sort(g[:, 2:5], -1) - np.sort(r[:, 2:5], -1)).max(-1) / lam
                print("  planes %3d-%3d: foreground %.3f, identical %.6f, max err %.3g, nan %d, ref nonfinite %d"
                      % (z, z + 63, float(mask[z:z + 64].mean()), float((got[z:z + 64] == ref[z:z + 64]).mean()),
                         float(np.nanmax(e)), int(np.isnan(e).sum()), int((~np.isfinite(r)).sum())), flush=True)
                if np.isnan(e).any():
                    i = int(np.flatnonzero(np.isnan(e))[0])
                    print("    first:", i, "got", got[z:z + 64].reshape(-1, 8)[i], "ref", ref[z:z + 64].reshape(-1, 8)[i], flush=True)
            del ref
        del got
