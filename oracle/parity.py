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
import itertools

import numpy as np


def eig_parity(got, ref, block=1 << 22, tie_tol=4e-6):
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
    and, about the voxels counted in order_diff (what such a voxel costs a consumer that reads
    Eigenvalue1..3 by position rather than as a set):
      order_max_elem_err   worst element-wise |got - ref|_inf / |lambda_1| among them (up to 2
                           when the tied pair has opposite signs: lambda and -lambda swap places)
      order_opposite_sign  how many of them tie with opposite signs
      order_max_tie_gap    largest | |a| - |b| | / |lambda_1| of a swapped pair, measured on the
                           REFERENCE triple: how far from an exact magnitude tie a swap occurred
      near_ties            voxels of the reference whose two closest magnitudes differ by at most
                           `tie_tol` * |lambda_1|: the only voxels where a rounding difference of
                           that size can change the order at all
    """
    C = got.shape[-1]
    e0 = C - 6 if C >= 6 else 0
    g2 = got.reshape(-1, C)
    r2 = ref.reshape(-1, C)
    out = {"max_err": 0.0, "order_diff": 0, "mag_slack": 0.0, "n": 0,
           "max_err_sum": 0.0, "max_err_frob": 0.0, "max_err_prod": 0.0,
           "order_max_elem_err": 0.0, "order_opposite_sign": 0, "order_max_tie_gap": 0.0, "near_ties": 0,
           "nonfinite": 0}
    for i in range(0, g2.shape[0], block):
        g = g2[i:i + block, e0:].astype(np.float64)
        r = r2[i:i + block, e0:].astype(np.float64)
        ok = np.isfinite(r[:, :3]).all(-1)
        if not ok.all():
            # a non-finite reference triple must be non-finite on the device as well
            if np.isfinite(g[~ok, :3]).all(-1).any():
                out["max_err"] = float("inf")
            g, r = g[ok], r[ok]
        # ... and a finite reference triple must be finite on the device: a NaN would otherwise
        # drop out of every max() below (max(x, nan) keeps x) and pass unseen
        gbad = ~np.isfinite(g[:, :3]).all(-1)
        if gbad.any():
            out["nonfinite"] += int(gbad.sum())
            out["max_err"] = float("inf")
            g, r = g[~gbad], r[~gbad]
        if g.shape[0] == 0:
            continue
        lam = np.maximum(np.abs(r[:, 0]), 1e-30)
        gs, rs = np.sort(g[:, :3], -1), np.sort(r[:, :3], -1)
        se = np.abs(gs - rs).max(-1) / lam
        de = np.abs(g[:, :3] - r[:, :3]).max(-1) / lam
        out["max_err"] = max(out["max_err"], float(se.max()))
        swapped = de > se
        out["order_diff"] += int(swapped.sum())
        ar = np.abs(r[:, :3])
        gap = np.minimum(ar[:, 0] - ar[:, 1], ar[:, 1] - ar[:, 2]) / lam  # >= 0 up to the reference's own slack
        out["near_ties"] += int((gap <= tie_tol).sum())
        if swapped.any():
            out["order_max_elem_err"] = max(out["order_max_elem_err"], float(de[swapped].max()))
            # which positions changed hands: the permutation of the reference triple that the
            # device triple matches best (identity first, so an exact fit is never "moved")
            rs_, gs_ = r[swapped, :3], g[swapped, :3]
            perms = np.array(list(itertools.permutations(range(3))))          # (6, 3)
            errs = np.abs(gs_[:, None, :] - rs_[:, perms]).max(-1)             # (m, 6)
            best = perms[np.argmin(errs, 1)]                                   # (m, 3)
            partner = np.take_along_axis(rs_, best, 1)                         # reference value now at each position
            moved = best != np.arange(3)[None, :]
            out["order_opposite_sign"] += int((moved & (rs_ * partner < 0)).any(-1).sum())
            gap = np.where(moved, np.abs(np.abs(rs_) - np.abs(partner)), 0.0).max(-1) / lam[swapped]
            out["order_max_tie_gap"] = max(out["order_max_tie_gap"], float(gap.max()))
        a = np.abs(g[:, :3])
        slack = np.maximum(np.maximum(a[:, 1] - a[:, 0], a[:, 2] - a[:, 1]), 0.0) / lam
        out["mag_slack"] = max(out["mag_slack"], float(slack.max()))
        out["n"] += g.shape[0]
        if g.shape[1] >= 6:
            # derived scalars: where the reference overflowed (a float32 product or sum of
            # squares can) the device must show the same non-finite value; elsewhere the error
            # is relative, and a non-finite device value is an infinite error
            def derived(k, slack=0.0):
                fin = np.isfinite(r[:, k])
                same = np.where(fin, True, (g[:, k] == r[:, k]) | (np.isnan(g[:, k]) & np.isnan(r[:, k])))
                if not same.all() or not np.isfinite(g[fin, k]).all():
                    return None, fin
                return np.maximum(np.abs(g[fin, k] - r[fin, k]) - slack, 0.0), fin
            d, fin = derived(3)
            out["max_err_sum"] = float("inf") if d is None else max(out["max_err_sum"], float((d / lam[fin]).max(initial=0.0)))
            # the product of three tiny eigenvalues lands among the float32 denormals, where one
            # unit in the last place (1.4e-45) is no longer small against lambda_1^3: two such
            # units are allowed before the relative bar applies
            d, fin = derived(4, 2.0 * 1.4012984643e-45)
            out["max_err_prod"] = float("inf") if d is None else max(out["max_err_prod"], float((d / lam[fin] ** 3).max(initial=0.0)))
            d, fin = derived(5)
            out["max_err_frob"] = float("inf") if d is None else max(out["max_err_frob"], float((d / lam[fin]).max(initial=0.0)))
    for k, v in out.items():  # nothing may have slipped through as NaN
        if isinstance(v, float) and v != v:
            out[k] = float("inf")
    return out


def assert_eig_parity(got, ref, tol, what="", max_order=None):
    """Assert the north_star style bar `tol` (relative to |lambda_1|) on triples and derived
    scalars; returns the measurement dict for reporting.

    Order: a triple may come out in another order only where the reference itself has a
    magnitude tie within 2 * tol (both sides' error): the number of such voxels (`near_ties`,
    counted on the reference) bounds the number of order differences, and a swapped pair must
    be such a tie (`order_max_tie_gap` <= 2 * tol).  `max_order` adds an absolute cap where a
    caller has measured one (the full-size test)."""
    p = eig_parity(got, ref, tie_tol=2 * tol)
    assert p["mag_slack"] <= tol, "%s: device triple out of magnitude order by %.3g" % (what, p["mag_slack"])
    assert p["max_err"] <= tol, "%s: eigenvalue error %.3g > %.3g" % (what, p["max_err"], tol)
    if got.shape[-1] >= 6:
        assert p["max_err_sum"] <= 2 * tol, "%s: sum error %.3g" % (what, p["max_err_sum"])
        assert p["max_err_frob"] <= tol, "%s: Frobenius error %.3g" % (what, p["max_err_frob"])
        assert p["max_err_prod"] <= 3 * tol, "%s: product error %.3g" % (what, p["max_err_prod"])
    assert p["order_diff"] <= p["near_ties"], \
        "%s: %d triples in another order but only %d magnitude ties within %.1e in the reference" % (
            what, p["order_diff"], p["near_ties"], 2 * tol)
    assert p["order_max_tie_gap"] <= 2 * tol, \
        "%s: a pair %.3g |lambda1| apart in magnitude changed places" % (what, p["order_max_tie_gap"])
    if max_order is not None:
        assert p["order_diff"] <= max_order, "%s: %d of %d triples in another order (cap %d)" % (
            what, p["order_diff"], p["n"], max_order)
    return p
