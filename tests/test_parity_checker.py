"""The checker checked: oracle/parity.py decides whether a device triple "equals" the
reference's, so its own rules are pinned here on hand-made cases (CPU, no library needed)."""
import numpy as np
import pytest

from oracle import parity


def triples(n=4096, seed=5):
    rng = np.random.default_rng(seed)
    t = rng.normal(size=(n, 3)).astype(np.float32) * 100
    idx = np.argsort(-np.abs(t), -1)
    t = np.take_along_axis(t, idx, 1)  # magnitude order, as the solver returns them
    out = np.zeros((n, 8), np.float32)
    out[:, 2:5] = t
    out[:, 5] = t.sum(-1)
    out[:, 6] = t.prod(-1)
    out[:, 7] = np.sqrt((t * t).sum(-1))
    return out


def test_identical_outputs_have_no_error():
    r = triples()
    p = parity.assert_eig_parity(r.copy(), r, 1e-6, "identical", max_order=0)
    assert p["max_err"] == 0 and p["order_diff"] == 0 and p["nonfinite"] == 0 and p["n"] == len(r)


def test_error_above_the_bar_fails():
    r = triples()
    g = r.copy()
    g[17, 3] *= np.float32(1 + 1e-4)
    with pytest.raises(AssertionError, match="eigenvalue error"):
        parity.assert_eig_parity(g, r, 1e-6, "perturbed")


def test_exact_magnitude_tie_may_swap_but_nothing_else():
    r = triples()
    r[3, 2:5] = (5.0, -5.0, 1.0)          # lambda and -lambda: a tie with opposite signs
    g = r.copy()
    g[3, 2:5] = (-5.0, 5.0, 1.0)
    p = parity.assert_eig_parity(g, r, 1e-6, "tie", max_order=1)
    assert p["order_diff"] == 1 and p["order_opposite_sign"] == 1 and p["order_max_tie_gap"] == 0
    with pytest.raises(AssertionError, match="cap"):
        parity.assert_eig_parity(g, r, 1e-6, "tie", max_order=0)
    # a pair that is NOT a tie changing places is an error, however the set compares
    r[9, 2:5] = (5.0, 3.0, 1.0)
    g[9, 2:5] = (3.0, 5.0, 1.0)
    with pytest.raises(AssertionError):
        parity.assert_eig_parity(g, r, 1e-6, "no tie")


def test_device_nan_where_the_reference_is_finite_fails():
    """max(x, nan) keeps x: a NaN must not drop out of the maxima unseen."""
    r = triples()
    for col in (2, 3, 4):
        g = r.copy()
        g[100, col] = np.nan
        p = parity.eig_parity(g, r)
        assert p["max_err"] == float("inf") and p["nonfinite"] == 1
        with pytest.raises(AssertionError):
            parity.assert_eig_parity(g, r, 1e-6, "nan")
    for col, key in ((5, "max_err_sum"), (6, "max_err_prod"), (7, "max_err_frob")):
        g = r.copy()
        g[200, col] = np.inf
        assert parity.eig_parity(g, r)[key] == float("inf")
        with pytest.raises(AssertionError):
            parity.assert_eig_parity(g, r, 1e-6, "inf")


def test_overflowed_derived_scalar_must_match_the_reference():
    r = triples()
    r[50, 6] = np.inf                      # the float32 product overflowed in the reference
    g = r.copy()
    parity.assert_eig_parity(g, r, 1e-6, "same overflow")
    g[50, 6] = -np.inf
    with pytest.raises(AssertionError):
        parity.assert_eig_parity(g, r, 1e-6, "other sign")
    r[60, 2:5] = np.nan                    # non-finite reference triple: the device must agree
    g = r.copy()
    g[50, 6] = np.inf
    parity.assert_eig_parity(g, r, 1e-6, "nan on both sides")
    g[60, 2:5] = 1.0
    with pytest.raises(AssertionError):
        parity.assert_eig_parity(g, r, 1e-6, "finite where the reference is not")
