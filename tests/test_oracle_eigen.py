"""CPU tests of the oracle's eigen solver (a1/a2) against the reference's own fixtures.

The seven cases of tests/golden/eigen_kat.json are the reference's known-answer tests
(test/Symmetric3x3EigenvalueSolverTest.cxx:48-90): EXPECT_FLOAT_EQ (4 float ULP) on a
double solver, EXPECT_NEAR 1e-15 for "Ones".  They are what pins this oracle.
"""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
KAT = json.load(open(os.path.join(HERE, "golden", "eigen_kat.json")))["cases"]


def float_eq(expected, actual):
    """gtest EXPECT_FLOAT_EQ: both rounded to float, at most 4 ULP apart."""
    e = np.float32(expected)
    a = np.float32(actual)
    if e == a:
        return True
    ei = np.array(e).view(np.int32).astype(np.int64)
    ai = np.array(a).view(np.int32).astype(np.int64)
    # biased representation as gtest does
    def biased(i):
        return np.where(i < 0, -(i & 0x7FFFFFFF), i)
    return bool(abs(int(biased(ei)) - int(biased(ai))) <= 4)


@pytest.mark.parametrize("case", KAT, ids=[c["name"] for c in KAT])
def test_reference_known_answers_double(oracle, case):
    ev = oracle.eig3(np.array(case["A"], np.float64))
    for k in range(3):
        if case["check"] == "near":
            assert abs(ev[k] - case["expected"][k]) <= case["epsilon"]
        else:
            assert float_eq(case["expected"][k], ev[k]), (case["name"], ev, case["expected"])


@pytest.mark.parametrize("trig", [0, 1])
@pytest.mark.parametrize("case", KAT, ids=[c["name"] for c in KAT])
def test_reference_known_answers_float_solver(oracle, case, trig):
    """TRealType=float (what every tool instantiates): within float resolution of ||A||."""
    ev = oracle.eig3(np.array(case["A"], np.float32), trig)
    exp = np.array(case["expected"])
    assert np.abs(ev - exp).max() <= 2e-6 * max(1.0, np.abs(exp).max())


def test_diagonal_branch_tie_rules(oracle):
    """Strict '>' tree of Symmetric3x3EigenvalueSolver.h:45-83: the else arm wins ties."""
    def run(d):
        return list(oracle.eig3(np.array([d[0], 0, 0, d[1], 0, d[2]], np.float32)))
    assert run([1, 1, 1]) == [1, 1, 1]
    assert run([1, -1, 0]) == [-1, 1, 0]          # |A11| > |A22| false -> A22 first
    assert run([-1, 1, 0]) == [1, -1, 0]
    assert run([2, 2, 1]) == [2, 2, 1]            # e0 = A22, e1 = A11
    assert run([3, 1, 3]) == [3, 3, 1]            # |A11| > |A33| false -> A33 first
    assert run([0, 0, 0]) == [0, 0, 0]
    # which entry comes first is observable with signs
    assert run([2, -2, 1]) == [-2, 2, 1]
    assert run([-3, 1, 3]) == [3, -3, 1]


def test_ordering_trace_and_against_numpy(oracle):
    rng = np.random.default_rng(11)
    A = rng.standard_normal((2000, 6))
    ev = oracle.eig3(A)
    a = np.abs(ev)
    assert (a[:, 0] >= a[:, 1]).all() and (a[:, 1] >= a[:, 2]).all()
    M = np.zeros((2000, 3, 3))
    M[:, 0, 0], M[:, 0, 1], M[:, 0, 2] = A[:, 0], A[:, 1], A[:, 2]
    M[:, 1, 1], M[:, 1, 2], M[:, 2, 2] = A[:, 3], A[:, 4], A[:, 5]
    M = M + np.transpose(M, (0, 2, 1)) - np.einsum("nij,ij->nij", M, np.eye(3))
    w = np.linalg.eigvalsh(M)
    np.testing.assert_allclose(np.sort(ev, axis=1), w, rtol=0, atol=1e-12 * np.abs(w).max())
    np.testing.assert_allclose(ev.sum(1), A[:, 0] + A[:, 3] + A[:, 5], atol=1e-12)


def test_float_solver_tracks_double_solver(oracle):
    rng = np.random.default_rng(12)
    A = (rng.standard_normal((5000, 6)) * 50).astype(np.float32)
    ev64 = oracle.eig3(A.astype(np.float64))
    for trig in (0, 1):
        ev32 = oracle.eig3(A, trig)
        scale = np.abs(ev64[:, :1])
        # float B/r arithmetic: error is absolute in ||A||, worst near double roots
        assert (np.abs(ev32 - ev64) / scale).max() < 2e-3
        assert np.median(np.abs(ev32 - ev64) / scale) < 2e-7
    d = np.abs(oracle.eig3(A, 0).astype(np.float64) - oracle.eig3(A, 1)) / np.abs(ev64[:, :1])
    assert d.max() < 5e-6  # the two include contexts (SURVEY TL;DR item 5)


def test_features_functor(oracle):
    rng = np.random.default_rng(13)
    A = rng.standard_normal((100, 6)).astype(np.float32)
    ev = oracle.eig3(A)
    f = oracle.eigfeat(A)
    np.testing.assert_array_equal(f[:, :3], ev)
    np.testing.assert_array_equal(f[:, 3], (ev[:, 0] + ev[:, 1]) + ev[:, 2])
    np.testing.assert_array_equal(f[:, 4], (ev[:, 0] * ev[:, 1]) * ev[:, 2])
    np.testing.assert_array_equal(
        f[:, 5], np.sqrt((ev[:, 0] * ev[:, 0] + ev[:, 1] * ev[:, 1]) + ev[:, 2] * ev[:, 2]))


def test_nan_propagates(oracle):
    ev = oracle.eig3(np.array([1, np.nan, 0, 1, 0, 1], np.float32))
    assert np.isnan(ev).all()


def test_self_pinned_eigen_fixture(oracle):
    z = np.load(os.path.join(HERE, "golden", "eigen_f32.npz"))
    np.testing.assert_array_equal(oracle.eig3(z["A"], 0), z["ev_cmath"])
    np.testing.assert_array_equal(oracle.eig3(z["A"], 1), z["ev_math_h"])
    np.testing.assert_array_equal(oracle.eigfeat(z["A"], 0), z["feat_cmath"])
