"""Reproduce one case of tests/test_gpu_fuzz.py in the default solver mode and dump the voxels
whose eigenvalue order differs from the oracle's (debugging aid)."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["IFE_TRIG_MODE"] = "0"
import test_gpu_fuzz as F
from oracle import pyoracle, parity
ife = importlib.import_module("image-feature-extraction_amd")
pyoracle.build(); pyoracle.set_threads(8)
seed = 20261004 + 2
rng = np.random.default_rng(seed)
want_case = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ctx = ife.Context(0); ctx.set_option(ife.OPT_TRIG_MODE, 2)
for case in range(want_case + 1):
    shape, spacing, sig = F._draw_case(rng)
    i16 = rng.random() < 0.3
    img = F._draw_volume(rng, shape, i16)
    mask = F._draw_mask(rng, shape)
    layout = ife.INTERLEAVED if rng.random() < 0.7 else ife.PLANAR
    if case != want_case:
        continue
    got = ctx.emphysema_features(img, mask, sig, spacing, layout)
    if layout == ife.PLANAR:
        got = np.moveaxis(got, 1, -1)
    omask = np.ones(shape, np.uint8) if mask is None else mask
    for s, sigma in enumerate(sig):
        ref = pyoracle.emphysema_features(img.astype(np.float32), omask, sigma, spacing)
        g, r = got[s].reshape(-1, 8).astype(np.float64), ref.reshape(-1, 8).astype(np.float64)
        p = parity.eig_parity(got[s], ref, tie_tol=2e-5)
        print("sigma", sigma, p)
        lam = np.maximum(np.abs(r[:, 2]), 1e-30)
        gs, rs = np.sort(g[:, 2:5], -1), np.sort(r[:, 2:5], -1)
        se = np.abs(gs - rs).max(-1) / lam
        de = np.abs(g[:, 2:5] - r[:, 2:5]).max(-1) / lam
        idx = np.nonzero(de > se)[0]
        for k in idx[:12]:
            print(k, "got", g[k, 2:5].tolist(), "ref", r[k, 2:5].tolist(), "se", se[k], "de", de[k])
