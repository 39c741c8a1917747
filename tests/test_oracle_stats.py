"""Rows f1 / f2 on the CPU: the oracle restatement of the reference's histogram-edge and
dense-histogram code against (1) the known answers of the reference's own tests
(test/DetermineEdgesForEqualizedHistogramTest.cxx:30-120, test/DenseHistogramTest.cxx:10-55),
(2) tests/golden/stats_ref.json, outputs of the reference's headers compiled as
oracle/_ref, and (3) that compiled reference itself where it is present."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def unhex(xs, dt=np.float32):
    return np.array([float.fromhex(x) for x in xs], dt)


@pytest.fixture(scope="module")
def golden():
    return json.load(open(os.path.join(HERE, "golden", "stats_ref.json")))


# ---- the reference's own known answers ----------------------------------------------------
def test_unique_equalizable(oracle):  # DetermineEdgesForEqualizedHistogramTest.cxx:30-39
    e = oracle.equalized_edges(np.arange(1, 10, dtype=np.float64), 3)
    assert e.tolist() == [4.0, 7.0]


def test_all_values_equal(oracle):  # :41-49
    e = oracle.equalized_edges(np.ones(8), 2)
    assert e.tolist() == [1.0]


def test_uneven_distribution(oracle):  # :51-60
    e = oracle.equalized_edges(np.array([1, 1, 1, 1, 1, 2, 2, 3, 3, 3], np.float64), 3)
    assert e.tolist() == [2.0, 3.0]


def test_too_many_bins(oracle):  # :62-70, std::out_of_range
    with pytest.raises(oracle.EdgeWalkError) as ei:
        oracle.equalized_edges(np.arange(1, 10, dtype=np.float64), 10)
    assert ei.value.rc == 1


def test_edges_are_increasing(oracle):  # :73-82 (seeded here; the reference seeds from random_device)
    v = np.sort(np.random.default_rng(1).uniform(-10, 10, 1000))
    e = oracle.equalized_edges(v, 50)
    assert e.size == 49 and np.all(np.diff(e) > 0)


def test_bins_are_equal_size(oracle):  # :84-120
    v = np.unique(np.random.default_rng(2).uniform(-10, 10, 1000))
    v = v[: v.size - v.size % 50]
    e = oracle.equalized_edges(v, 50)
    # a bin is [e_i, e_i+1) in this test of the reference
    counts = np.diff(np.concatenate([[0], np.searchsorted(v, e, side="left"), [v.size]]))
    assert np.all(counts == v.size // 50)


def test_dense_histogram_counts_and_frequencies(oracle):  # DenseHistogramTest.cxx:10-55
    vals = [-1, 0, 0.5, 1, 1.5, 2.1, 2.6, 2.9, 3.2, 3.5, 4.2, 4.6, 5, 6, 7, 8, 9, 10]
    c, f = oracle.dense_histogram([1, 2.5, 3.0, 4.7, 6.2, 8.3], vals)
    assert c.tolist() == [4, 2, 2, 4, 2, 2, 2]
    np.testing.assert_allclose(f, np.array([4, 2, 2, 4, 2, 2, 2], np.float32) / 18, rtol=4 * 1.2e-7)


# ---- fixtures generated from the compiled reference -------------------------------------------
def test_edges_match_reference_fixtures(oracle, golden):
    for c in golden["edges"]:
        v32 = unhex(c["values"])
        if "error" in c:
            with pytest.raises(oracle.EdgeWalkError):
                oracle.equalized_edges(v32, c["nbins"])
            continue
        assert np.array_equal(oracle.equalized_edges(v32, c["nbins"]), unhex(c["edges_f32"])), c["kind"]
        assert np.array_equal(oracle.equalized_edges(v32.astype(np.float64), c["nbins"]),
                              unhex(c["edges_f64"], np.float64)), c["kind"]


def test_dense_histogram_matches_reference_fixtures(oracle, golden):
    for h in golden["dense_histogram"]:
        c, f = oracle.dense_histogram(unhex(h["edges"]), unhex(h["values"]))
        assert c.tolist() == h["counts"]
        assert np.array_equal(f, unhex(h["freqs"]))


def test_sort_and_gather(oracle):
    rng = np.random.default_rng(3)
    v = rng.normal(0, 5, 1000).astype(np.float32)
    assert np.array_equal(oracle.sort_f32(v), np.sort(v))
    feat = rng.normal(0, 1, (4, 5, 6, 8)).astype(np.float32)
    mask = rng.integers(0, 3, (4, 5, 6)).astype(np.uint8)
    cols = oracle.gather_foreground(feat, mask, [2, 1])
    ref = feat.reshape(-1, 8)[(mask.ravel() == 1) | (mask.ravel() == 2)].T
    assert np.array_equal(cols, ref)
    assert oracle.gather_foreground(feat, mask, [7]).shape == (8, 0)


# ---- the compiled reference itself (this container; the .so travels to the GPU box) ------
def test_restatement_against_compiled_reference(oracle):
    if oracle.ref_lib() is None:
        pytest.skip("oracle/_ref not built and /root/reference absent")
    rng = np.random.default_rng(4)
    checked = walked = 0
    for _ in range(2000):
        n = int(rng.integers(1, 600))
        nb = int(rng.integers(1, 70))
        span = [4, 40, 10 ** 6][int(rng.integers(0, 3))]
        v = np.sort(rng.integers(-span, span, n).astype(np.float32) / 4)
        try:
            a = oracle.equalized_edges(v, nb)
        except oracle.EdgeWalkError as e:
            if e.rc == 1:
                with pytest.raises(oracle.EdgeWalkError):
                    oracle.ref_equalized_edges(v, nb)
            walked += e.rc == 3  # the reference asserts here: not called
            continue
        assert np.array_equal(a, oracle.ref_equalized_edges(v, nb))
        checked += 1
    assert checked > 1000
    for _ in range(50):
        edges = np.unique(rng.normal(0, 3, int(rng.integers(1, 60))).astype(np.float32))
        vals = np.round(rng.normal(0, 4, 500).astype(np.float32), 1)
        c, f = oracle.dense_histogram(edges, vals)
        rc, rf = oracle.ref_dense_histogram(edges, vals)
        assert np.array_equal(c, rc) and np.array_equal(f, rf)


def test_pair_list_and_text_format_of_the_reference(oracle, golden, tmp_path):
    """Formats at the tool boundary (IO.h:24-41, src/IO/IO.cxx:20-41), as fixtures; the host
    mirror is held to the same strings in test_host_tools.py."""
    for t in golden["write_sequence"]:
        assert isinstance(t["text"], str) and "," in t["text"] or len(t["values"]) == 1
    if oracle.ref_lib() is None:
        pytest.skip("oracle/_ref not built and /root/reference absent")
    p = tmp_path / "pairs.csv"
    p.write_text("a.nii.gz, m a.nii.gz \n\n b.nii,b_mask.nii\r\n")
    assert oracle.ref_read_pair_list(str(p)) == [("a.nii.gz", "m a.nii.gz"), ("b.nii", "b_mask.nii")]
    for t in golden["write_sequence"]:
        assert oracle.ref_write_sequence(unhex(t["values"])) == t["text"]
