"""Comparison helpers shared by tests/, __graft_entry__.smoke() and bench.py's checker leg.

Test infrastructure (like everything under oracle/): nothing in the product imports it.

The eigenvalue triple of Symmetric3x3EigenvalueSolver.h:123-129 is ordered by MAGNITUDE
(|e0| >= |e1| >= |e2|), which is discontinuous where two magnitudes tie: two evaluations
that differ by one rounding may legitimately return the tied pair in either order (the
reference's own two include contexts, double and float trig, do).  Parity of the triple is
therefore measured on the triples sorted by VALUE -- any real error shows there -- plus the
requirement that each side is magnitude-ordered; voxels whose order differs are counted and
reported.  All errors are relative to |lambda_1| of the reference (north_star: 1e-5).
"""
import numpy as np


def eig_parity(got, ref, block=1 << 22):
    """got, ref: (..., C) with the eigenvalue triple in the first 3 of the last 6 columns
    ([e0, e1, e2, sum, product, frobenius]; C = 3, 6 or 8).  Returns a dict:
      max_err      max over voxels of |sorted(got) - sorted(ref)|_inf / |lambda_1|
      max_err_sum, max_err_frob  (/|lambda_1|), max_err_prod (/|lambda_1|^3)  when C >= 6
      order_diff   number of voxels whose triple is the same set in another order
      mag_slack    max over voxels of (|e1| - |e0|, |e2| - |e1|, 0) / |lambda_1| of the device
                   triple: 0 when it is magnitude-ordered.  The reference's own order is only
                   as good as its float rounding (e1 = 3q - e0 - e2 can pass a nearly equal e0
                   by an ulp after the :123-129 swaps), so this is a tolerance, not an identity
      n            voxels compared (finite reference)
    """
    C = got.shape[-1]
    e0 = C - 6 if C >= 6 else 0
    g2 = got.reshape(-1, C)
    r2 = ref.reshape(-1, C)
    out = {"max_err": 0.0, "order_diff": 0, "mag_slack": 0.0, "n": 0,
           "max_err_sum": 0.0, "max_err_frob": 0.0, "max_err_prod": 0.0}
    for i in range(0, g2.shape[0], block):
        g = g2[i:i + block, e0:].astype(np.float64)
        r = r2[i:i + block, e0:].astype(np.float64)
        ok = np.isfinite(r[:, :3]).all(-1)
        if not ok.all():
            # a non-finite reference triple must be non-finite on the device as well
            if np.isfinite(g[~ok, :3]).all(-1).any():
                out["max_err"] = float("inf")
            g, r = g[ok], r[ok]
        if g.shape[0] == 0:
            continue
        lam = np.maximum(np.abs(r[:, 0]), 1e-30)
        gs, rs = np.sort(g[:, :3], -1), np.sort(r[:, :3], -1)
        se = np.abs(gs - rs).max(-1) / lam
        de = np.abs(g[:, :3] - r[:, :3]).max(-1) / lam
        out["max_err"] = max(out["max_err"], float(se.max()))
        out["order_diff"] += int((de > se).sum())
        a = np.abs(g[:, :3])
        slack = np.maximum(np.maximum(a[:, 1] - a[:, 0], a[:, 2] - a[:, 1]), 0.0) / lam
        out["mag_slack"] = max(out["mag_slack"], float(slack.max()))
        out["n"] += g.shape[0]
        if g.shape[1] >= 6:
            out["max_err_sum"] = max(out["max_err_sum"], float((np.abs(g[:, 3] - r[:, 3]) / lam).max()))
            # the product of three tiny eigenvalues lands among the float32 denormals, where one
            # unit in the last place (1.4e-45) is no longer small against lambda_1^3: two such
            # units are allowed before the relative bar applies
            dp = np.maximum(np.abs(g[:, 4] - r[:, 4]) - 2.0 * 1.4012984643e-45, 0.0)
            out["max_err_prod"] = max(out["max_err_prod"], float((dp / lam ** 3).max()))
            out["max_err_frob"] = max(out["max_err_frob"], float((np.abs(g[:, 5] - r[:, 5]) / lam).max()))
    return out


def assert_eig_parity(got, ref, tol, what="", max_order_frac=1e-4):
    """Assert the north_star style bar `tol` (relative to |lambda_1|) on triples and derived
    scalars; returns the measurement dict for reporting."""
    p = eig_parity(got, ref)
    assert p["mag_slack"] <= tol, "%s: device triple out of magnitude order by %.3g" % (what, p["mag_slack"])
    assert p["max_err"] <= tol, "%s: eigenvalue error %.3g > %.3g" % (what, p["max_err"], tol)
    if got.shape[-1] >= 6:
        assert p["max_err_sum"] <= 2 * tol, "%s: sum error %.3g" % (what, p["max_err_sum"])
        assert p["max_err_frob"] <= tol, "%s: Frobenius error %.3g" % (what, p["max_err_frob"])
        assert p["max_err_prod"] <= 3 * tol, "%s: product error %.3g" % (what, p["max_err_prod"])
    assert p["order_diff"] <= max(2, max_order_frac * p["n"]), \
        "%s: %d of %d triples in another order" % (what, p["order_diff"], p["n"])
    return p
