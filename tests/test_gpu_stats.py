"""Rows f1 / f2 on the GPU, through the C-ABI: device radix sort, equalizing edges, dense
histogram and the sample columns of DetermineHistogramBinEdges_MultiScaleEigenvalueFeatures,
against the oracle, the reference-generated fixtures (tests/golden/stats_ref.json) and,
where it travelled with the snapshot, the compiled reference (oracle/_ref).  Bit-exact."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def unhex(xs, dt=np.float32):
    return np.array([float.fromhex(x) for x in xs], dt)


def same_multiset_sorted(out, src):
    assert np.array_equal(out, np.sort(src))            # values ascending (-0 == +0 here)
    assert np.array_equal(np.sort(out.view(np.uint32)), np.sort(src.view(np.uint32)))  # same bits


@pytest.mark.parametrize("n", [0, 1, 2, 63, 64, 65, 1023, 4095, 4096, 4097, 8193, 100003, 1 << 20])
def test_sort_sizes(ctx, n):
    rng = np.random.default_rng(n + 1)
    v = (rng.normal(0, 1000, n) * rng.choice([1e-30, 1.0, 1e20], n)).astype(np.float32)
    same_multiset_sorted(ctx.sort_f32(v), v)


def test_sort_special_values(ctx):
    rng = np.random.default_rng(7)
    v = np.concatenate([
        np.array([0.0, -0.0, np.inf, -np.inf, 1e-45, -1e-45, 3.4e38, -3.4e38], np.float32),
        np.zeros(5000, np.float32), -np.zeros(3000, np.float32),
        rng.integers(-3, 4, 20000).astype(np.float32),
        np.full(9000, 2.5, np.float32)])
    rng.shuffle(v)
    out = ctx.sort_f32(v)
    same_multiset_sorted(out, v)
    z = out[out == 0]
    assert np.all(np.diff(np.signbit(z).astype(int)) <= 0)  # every -0 before every +0


def test_sort_all_equal_and_presorted(ctx):
    v = np.full(70001, -7.25, np.float32)
    assert np.array_equal(ctx.sort_f32(v), v)
    a = np.arange(50000, dtype=np.float32)
    assert np.array_equal(ctx.sort_f32(a), a)
    assert np.array_equal(ctx.sort_f32(a[::-1].copy()), a)


def test_sort_large(ctx):
    n = (1 << 26) + 5
    rng = np.random.default_rng(11)
    v = rng.standard_normal(n, dtype=np.float32)
    v[::7] = np.round(v[::7])  # many duplicates
    out = ctx.sort_f32(v)
    assert np.all(out[1:] >= out[:-1])
    assert np.array_equal(out, np.sort(v))


def test_edges_reference_known_answers(ctx, ife):
    # test/DetermineEdgesForEqualizedHistogramTest.cxx:30-70
    for dt in (np.float32, np.float64):  # the reference's test runs the template on double
        assert ctx.equalized_edges(np.arange(1, 10, dtype=dt), 3).tolist() == [4.0, 7.0]
        assert ctx.equalized_edges(np.ones(8, dt), 2).tolist() == [1.0]
        assert ctx.equalized_edges(np.array([1, 1, 1, 1, 1, 2, 2, 3, 3, 3], dt), 3).tolist() == [2.0, 3.0]
    with pytest.raises(ife.IfeError) as ei:
        ctx.equalized_edges(np.arange(1, 10), 10)
    assert ei.value.code == ife.E_ARG and "Too many bins" in str(ei.value)
    assert ctx.equalized_edges(np.arange(5), 1).size == 0


def test_edges_reference_fixtures(ctx, ife):
    g = json.load(open(os.path.join(HERE, "golden", "stats_ref.json")))
    for c in g["edges"]:
        v = unhex(c["values"])
        if "error" in c:
            with pytest.raises(ife.IfeError):
                ctx.equalized_edges(v, c["nbins"])
            continue
        assert np.array_equal(ctx.equalized_edges(v, c["nbins"]), unhex(c["edges_f32"])), c["kind"]
        assert np.array_equal(ctx.equalized_edges(v.astype(np.float64), c["nbins"]),
                              unhex(c["edges_f64"], np.float64)), c["kind"]


def test_edges_random_against_oracle_and_compiled_reference(ctx, ife, oracle):
    rng = np.random.default_rng(5)
    ref = oracle.ref_lib()
    ok = walked = 0
    for _ in range(300):
        n = int(rng.integers(1, 3000))
        nb = int(rng.integers(1, 80))
        span = [4, 40, 10 ** 6][int(rng.integers(0, 3))]
        v = np.sort(rng.integers(-span, span, n).astype(np.float32) / 4)
        try:
            want = oracle.equalized_edges(v, nb)
        except oracle.EdgeWalkError as e:
            with pytest.raises(ife.IfeError) as ei:
                ctx.equalized_edges(v, nb)
            assert ei.value.code == (ife.E_ARG if e.rc == 1 else ife.E_STATE)
            walked += 1
            continue
        got = ctx.equalized_edges(v, nb)
        assert np.array_equal(got, want)
        if ref is not None:
            assert np.array_equal(got, oracle.ref_equalized_edges(v, nb))
        ok += 1
    assert ok > 150


def test_dense_histogram(ctx, oracle):
    g = json.load(open(os.path.join(HERE, "golden", "stats_ref.json")))
    for h in g["dense_histogram"]:
        assert ctx.dense_histogram(unhex(h["edges"]), unhex(h["values"])).tolist() == h["counts"]
    vals = [-1, 0, 0.5, 1, 1.5, 2.1, 2.6, 2.9, 3.2, 3.5, 4.2, 4.6, 5, 6, 7, 8, 9, 10]
    assert ctx.dense_histogram([1, 2.5, 3.0, 4.7, 6.2, 8.3], vals).tolist() == [4, 2, 2, 4, 2, 2, 2]
    rng = np.random.default_rng(9)
    edges = np.unique(rng.normal(0, 3, 1000).astype(np.float32))
    v = np.round(rng.normal(0, 4, 2_000_003).astype(np.float32), 2)
    c, _ = oracle.dense_histogram(edges, v)
    assert np.array_equal(ctx.dense_histogram(edges, v), c)


def labels(shape, seed):
    rng = np.random.default_rng(seed)
    m = rng.integers(0, 3, shape).astype(np.uint8)
    m[: shape[0] // 4] = 0
    return m


@pytest.mark.parametrize("fg", [(1,), (2,), (1, 2), (2, 1, 5)])
def test_samples_add_features_foreground(ctx, oracle, fg):
    rng = np.random.default_rng(21)
    shape = (9, 10, 11)
    feat = rng.normal(0, 3, shape + (8,)).astype(np.float32)
    m = labels(shape, 22)
    s = ctx.samples(16)
    s.add_features(8, feat, mask=m, foreground=fg)
    want = oracle.gather_foreground(feat, m, fg)
    assert s.count(8) == want.shape[1] and s.count(0) == 0
    for c in range(8):
        assert np.array_equal(s.column(8 + c), want[c])  # raster order, as the tool pushes them
    s.sort()
    for c in range(8):
        assert np.array_equal(s.column(8 + c), np.sort(want[c]))
    s.close()


def test_samples_add_features_indexed_and_planar(ctx, ife):
    rng = np.random.default_rng(23)
    shape = (6, 7, 8)
    feat = rng.normal(0, 3, (8,) + shape).astype(np.float32)
    idx = rng.integers(0, 6 * 7 * 8, 1000)
    s = ctx.samples(8)
    s.add_features(0, feat, indices=idx, layout=ife.PLANAR)
    s.add_features(0, feat, indices=idx[:10], layout=ife.PLANAR)
    assert s.count(3) == 1010
    for c in range(8):
        col = feat[c].ravel()
        assert np.array_equal(s.column(c), np.concatenate([col[idx], col[idx[:10]]]))
    s.close()


def test_samples_pipeline_matches_tool_semantics(ctx, oracle, synth):
    """Two images, two scales, labels {0,1,2}, foreground {1,2}: clamp -> a5 -> gather ->
    sort -> edges, against the oracle doing the same on the CPU (tool :147-289)."""
    sigmas = [1.0, 2.5]
    s = ctx.samples(16)
    cols = [[] for _ in range(16)]
    for k, shape in enumerate([(20, 24, 28), (16, 30, 22)]):
        img = synth.volume_f32(shape, 100 + k)
        m = labels(shape, 30 + k)
        s.add_image(img, m, sigmas, foreground=(1, 2))
        clamped = np.minimum(m, 1).astype(np.uint8)
        for i, sg in enumerate(sigmas):
            f = oracle.emphysema_features(img, clamped, sg)
            g = oracle.gather_foreground(f, m, (1, 2))
            for c in range(8):
                cols[i * 8 + c].append(g[c])
    n = sum(x.size for x in cols[0])
    assert all(s.count(c) == n for c in range(16))
    got = s.equalized_edges(41)
    for c in range(16):
        v = oracle.sort_f32(np.concatenate(cols[c]))
        assert np.array_equal(s.column(c), v), c
        assert np.array_equal(got[c], oracle.equalized_edges(v, 41)), c
    s.clear()
    assert s.count(5) == 0
    s.close()


def test_samples_errors(ctx, ife):
    s = ctx.samples(8)
    feat = np.zeros((4, 4, 4, 8), np.float32)
    m = np.ones((4, 4, 4), np.uint8)
    with pytest.raises(ife.IfeError):
        s.add_features(4, feat, mask=m)            # columns 4..11 out of range
    with pytest.raises(ife.IfeError):
        s.add_features(0, feat, mask=m, foreground=())
    with pytest.raises(ife.IfeError):
        s.add_features(0, feat, indices=[64])      # voxel index out of range
    s.add_features(0, feat, mask=m)
    with pytest.raises(ife.IfeError) as ei:
        s.equalized_edges(65)                       # 64 samples, 65 bins
    assert "Too many bins" in str(ei.value)
    assert np.array_equal(s.equalized_edges(4), np.zeros((8, 3), np.float32))
    with pytest.raises(ife.IfeError):
        s.add_image(np.zeros((4, 4, 4), np.float32), m, [1.0, 2.0])  # needs 16 columns
    s.close()


# ---- row f2: the bag rows of MakeBag (tools/MakeBag.cxx:405-472) -------------------------------
def random_rois(rng, shape, n, size):
    nz, ny, nx = shape
    sx, sy, sz = size
    return np.stack([rng.integers(0, nx - sx + 1, n), rng.integers(0, ny - sy + 1, n),
                     rng.integers(0, nz - sz + 1, n), np.full(n, sx), np.full(n, sy), np.full(n, sz)], 1)


def test_roi_histograms_against_oracle(ctx, ife, oracle):
    rng = np.random.default_rng(41)
    shape = (18, 22, 26)
    feat = np.round(rng.normal(0, 3, shape + (8,)), 1).astype(np.float32)  # values on edges happen
    m = labels(shape, 42)
    edges = np.sort(np.round(rng.normal(0, 3, (8, 13)), 1).astype(np.float32), axis=1)
    rois = np.concatenate([random_rois(rng, shape, 9, (7, 5, 6)), [[0, 0, 0, 26, 22, 18]],
                           [[25, 21, 17, 1, 1, 1]], [[3, 3, 0, 5, 5, 4]]])  # whole volume, one voxel, empty mask
    want, freqs = oracle.roi_histograms(feat, np.minimum(m, 1), rois, edges)
    got = ctx.roi_histograms(feat, m, rois, edges)
    assert np.array_equal(got, want)
    assert want[-1].sum() == 0 and np.isnan(freqs[-1]).all()      # 0/0 as in getFrequencies
    planar = np.ascontiguousarray(np.moveaxis(feat, -1, 0))
    assert np.array_equal(ctx.roi_histograms(planar, m.astype(np.uint16), rois, edges, layout=ife.PLANAR), want)
    with pytest.raises(ife.IfeError) as ei:
        ctx.roi_histograms(feat, m, [[20, 0, 0, 7, 5, 6]], edges)
    assert ei.value.code == ife.E_ARG and "outside of the largest possible region" in str(ei.value)


def test_bag_image_matches_tool_semantics(ctx, oracle, synth):
    rng = np.random.default_rng(43)
    shape = (24, 28, 32)
    img = synth.volume_f32(shape, 77)
    lab = synth.mask_ellipsoids(shape)
    sigmas = [1.0, 2.0]
    clamped = np.minimum(lab, 1).astype(np.uint8)
    feats = [oracle.emphysema_features(img, clamped, s) for s in sigmas]
    # histogram specification: equalizing edges of the whole foreground, 9 bins
    edges = np.stack([oracle.equalized_edges(oracle.sort_f32(f[..., c][clamped != 0]), 9)
                      for f in feats for c in range(8)])
    rois = random_rois(rng, shape, 12, (9, 9, 7))
    got = ctx.bag_image(img, lab, sigmas, rois, edges)
    assert got.shape == (12, 16, 9)
    for i, f in enumerate(feats):
        want, _ = oracle.roi_histograms(f, clamped, rois, edges[i * 8:(i + 1) * 8])
        assert np.array_equal(got[:, i * 8:(i + 1) * 8, :], want), i


def test_samples_add_image_u16_labels_and_spacing(ctx, oracle, synth):
    """Fused sampling with uint16 labels, anisotropic spacing (the generic stencil path), a
    ragged width (the last row segment is partial) and a label set that includes 0."""
    shape, spacing, sigmas = (12, 18, 70), (0.7, 0.9, 1.6), [1.5]
    img = synth.volume_f32(shape, 55)
    lab = labels(shape, 56).astype(np.uint16) * 300            # labels 0, 300, 600
    clamped = np.minimum(lab, 1).astype(np.uint8)
    f = oracle.emphysema_features(img, clamped, sigmas[0], spacing)
    for fg in [(600,), (0, 300)]:
        s = ctx.samples(8)
        s.add_image(img, lab, sigmas, foreground=fg, spacing=spacing)
        sel = np.isin(lab, fg)
        assert s.count(0) == int(sel.sum())
        for c in range(8):
            want = f[..., c][sel]                                 # raster order
            got = s.column(c)
            if c < 2:
                assert np.array_equal(got, want), (fg, c)
            else:                                                 # eigen features: last-bit freedom only
                lam = np.maximum(np.abs(f[..., 2][sel]).astype(np.float64), 1e-30) ** (3 if c == 6 else 1)
                assert (np.abs(got.astype(np.float64) - want) / lam).max() <= 3e-6, (fg, c)
        if 0 in fg:                                               # label 0 is sampled: features are zero there
            zero = (lab == 0)[sel]
            assert np.all(s.column(3)[zero] == 0)
        s.close()
