"""The C++ host mirror: tool surface (flags, output names, exit codes) and results.

CPU part: the tools build, print usage, reject bad command lines with EXIT_FAILURE and --
there being no GPU here -- fail loudly instead of falling back to a CPU path.
GPU part (-m gpu): the tools produce the reference's output file names and the oracle's
values; host_selftest exercises the filter classes and functors.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "image-feature-extraction_amd", "host")
BIN = os.path.join(HOST, "bin")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import niftiio  # noqa: E402

TOOLS = ["ExtractFeatures", "FiniteDifference_HessianFeatures", "FiniteDifference_GradientFeatures",
         "MaskedNormalizedConvolution", "MaskedImageFilter"]


@pytest.fixture(scope="module")
def built():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "image-feature-extraction_amd", "csrc")])
    subprocess.check_call(["make", "-s", "-C", HOST])
    return BIN


def run(tool, *args):
    return subprocess.run([os.path.join(BIN, tool)] + list(args), capture_output=True, text=True)


@pytest.mark.parametrize("tool", TOOLS)
def test_usage_and_argument_errors(built, tool):
    r = run(tool, "--help")
    assert r.returncode == 0 and "--image" in r.stdout and "USAGE" in r.stdout
    r = run(tool)  # required arguments missing: TCLAP-style message, EXIT_FAILURE
    assert r.returncode == 1 and "Error :" in r.stderr and "for arg" in r.stderr
    r = run(tool, "--bogus", "1")
    assert r.returncode == 1 and "Couldn't find match" in r.stderr


def test_edges_tool_usage(built):
    tool = "DetermineHistogramBinEdges_MultiScaleEigenvalueFeatures"
    h = run(tool, "--help")
    assert h.returncode == 0
    assert all(f in h.stdout for f in ("--infile", "--outfile", "--bins", "--samples", "--scale", "--foreground"))
    r = run(tool, "-i", "x", "-o", "y")     # -b -S -s -f are required in the reference too
    assert r.returncode == 1 and "Error :" in r.stderr


def test_reference_flag_names(built):
    assert all(f in run("ExtractFeatures", "--help").stdout for f in ("-i,", "-m,", "-o,", "-s,", "--scale"))
    h = run("FiniteDifference_HessianFeatures", "--help").stdout
    assert "--outdir" in h and "--prefix" in h
    h = run("MaskedNormalizedConvolution", "--help").stdout
    assert "--certainty" in h and "--maskoutput" in h and "--scale" in h
    assert "--outside-value" in run("MaskedImageFilter", "--help").stdout


def test_no_gpu_means_failure_not_fallback(built, tmp_path, synth):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    img = synth.volume_f32((8, 8, 8), 1)
    niftiio.write(str(tmp_path / "i.nii"), img)
    niftiio.write(str(tmp_path / "m.nii"), np.ones((8, 8, 8), np.uint8))
    r = run("ExtractFeatures", "-i", str(tmp_path / "i.nii"), "-m", str(tmp_path / "m.nii"),
            "-o", str(tmp_path / "o"), "-s", "1")
    assert r.returncode == 1
    assert "Failed to process." in r.stderr and "no CPU path" in r.stderr


def test_missing_input_file(built, tmp_path):
    r = run("MaskedImageFilter", "-i", str(tmp_path / "nope.nii"), "-m", str(tmp_path / "nope.nii"),
            "-o", str(tmp_path / "o.nii"))
    assert r.returncode == 1 and "cannot open" in r.stderr


@pytest.mark.parametrize("what,patch,msg", [
    ("negative dim", [("<h", 42, -5)], "non-positive dimension"),
    ("zero dim", [("<h", 46, 0)], "non-positive dimension"),
    ("nan vox_offset", [("<f", 108, float("nan"))], "bad vox_offset"),
    ("huge vox_offset", [("<f", 108, 3.0e38)], "bad vox_offset"),
    ("vox_offset inside the header", [("<f", 108, 100.0)], "bad vox_offset"),
    ("more voxels than bytes", [("<h", 44, 3000)], "truncated voxel data"),
])
def test_malformed_nifti_headers_are_refused(built, tmp_path, what, patch, msg):
    """Header fields of an untrusted file are checked before they size anything."""
    niftiio.write(str(tmp_path / "bad.nii"), np.zeros((4, 5, 6), np.float32), patch=patch)
    niftiio.write(str(tmp_path / "ok.nii"), np.zeros((4, 5, 6), np.float32))
    r = run("MaskedImageFilter", "-i", str(tmp_path / "bad.nii"), "-m", str(tmp_path / "ok.nii"),
            "-o", str(tmp_path / "o.nii"))
    assert r.returncode == 1 and msg in r.stderr, (what, r.stderr)


def test_malformed_metaimage_is_refused(built, tmp_path):
    (tmp_path / "v.raw").write_bytes(b"\0" * 16)
    for dims in ("0 4 4", "4 4", "99999999999 4 4"):
        (tmp_path / "v.mhd").write_text("NDims = 3\nDimSize = %s\nElementType = MET_FLOAT\n"
                                        "ElementDataFile = v.raw\n" % dims)
        r = run("MaskedImageFilter", "-i", str(tmp_path / "v.mhd"), "-m", str(tmp_path / "v.mhd"),
                "-o", str(tmp_path / "o.nii"))
        assert r.returncode == 1 and ("DimSize" in r.stderr or "too large" in r.stderr
                                      or "truncated" in r.stderr), r.stderr


def test_host_tools_under_sanitizers(tmp_path, synth):
    """AddressSanitizer + UBSan build of the host side (`make SANITIZE=1 BINDIR=bin_san`),
    run over the paths that need no GPU: usage, argument errors, file reading, malformed
    headers, ROI generation, and the loud failure where the device would be needed."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "image-feature-extraction_amd", "csrc")])
    subprocess.check_call(["make", "-s", "-j", "8", "-C", HOST, "SANITIZE=1", "BINDIR=bin_san"])
    san = os.path.join(HOST, "bin_san")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    niftiio.write(str(tmp_path / "i.nii.gz"), synth.volume_f32((8, 9, 10), 1))
    niftiio.write(str(tmp_path / "m.nii.gz"), np.ones((8, 9, 10), np.uint16))
    niftiio.write(str(tmp_path / "bad.nii"), np.zeros((4, 5, 6), np.float32), patch=[("<f", 108, float("nan"))])
    (tmp_path / "rois.txt").write_text("index size\n[1, 1, 1][3, 3, 3]\n")
    (tmp_path / "h.txt").write_text("# c\n0,1,2\n")
    i, m = str(tmp_path / "i.nii.gz"), str(tmp_path / "m.nii.gz")
    cases = [["ExtractFeatures", "--help"], ["MakeBagDense", "--help"], ["MakeBag", "-i", "x"],
             ["ExtractFeatures", "-i", i, "-m", m, "-o", str(tmp_path / "o"), "-s", "1", "-s", "2"],
             ["MaskedImageFilter", "-i", str(tmp_path / "bad.nii"), "-m", m, "-o", str(tmp_path / "o.nii")],
             ["MakeBagOnlyIntensity", "-i", i, "-m", m, "-H", str(tmp_path / "h.txt"), "-o", str(tmp_path),
              "-r", str(tmp_path / "rois.txt")],
             ["MakeBagDense", "-i", i, "-m", m, "-H", str(tmp_path / "h.txt"), "-o", str(tmp_path), "-s", "1",
              "-x", "3", "-y", "3", "-z", "3"],
             ["DetermineHistogramBinEdges_MultiScaleEigenvalueFeatures", "-i", str(tmp_path / "nolist.txt"),
              "-o", str(tmp_path / "e.txt"), "-b", "4", "-S", "0", "-s", "1", "-f", "1"]]
    for c in cases:
        r = subprocess.run([os.path.join(san, c[0])] + c[1:], capture_output=True, text=True, env=env)
        assert "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, (c, r.stderr[-1500:])
        assert r.returncode in (0, 1), (c, r.returncode, r.stderr[-500:])


@pytest.mark.gpu
def test_world_geometry_survives_the_tools(built, tmp_path, synth):
    """A volume written by ITK carries an LPS qform/sform (quatern_d = 1, negated x/y
    offsets).  The tools compute with the spacing only, but their outputs must overlay the
    source: every geometry field of the header comes back as it went in."""
    shape = (8, 9, 10)
    geom = {"qform_code": (1,), "sform_code": (1,), "quatern": (0.0, 0.0, 1.0), "qoffset": (12.5, -7.25, 33.0),
            "srow_x": (-0.7, 0.0, 0.0, 12.5), "srow_y": (0.0, -0.8, 0.0, -7.25), "srow_z": (0.0, 0.0, 1.25, 33.0),
            "qfac": (1.0,)}
    niftiio.write(str(tmp_path / "i.nii.gz"), synth.volume_f32(shape, 3), (0.7, 0.8, 1.25), geometry=geom)
    niftiio.write(str(tmp_path / "m.nii.gz"), np.ones(shape, np.uint8), (0.7, 0.8, 1.25), geometry=geom)
    want = niftiio.read_geometry(str(tmp_path / "i.nii.gz"))
    r = run("ExtractFeatures", "-i", str(tmp_path / "i.nii.gz"), "-m", str(tmp_path / "m.nii.gz"),
            "-o", str(tmp_path / "o"), "-s", "1")
    assert r.returncode == 0, r.stderr
    got = niftiio.read_geometry(str(tmp_path / ("o_scale_1.000000GaussianBlur.nii.gz")))
    assert got == want
    r = run("MaskedImageFilter", "-i", str(tmp_path / "i.nii.gz"), "-m", str(tmp_path / "m.nii.gz"),
            "-o", str(tmp_path / "masked.nii"))
    assert r.returncode == 0, r.stderr
    assert niftiio.read_geometry(str(tmp_path / "masked.nii")) == want


@pytest.mark.gpu
def test_extract_features_over_several_devices(built, tmp_path, synth):
    """IFE_DEVICES: the tool cuts the volume into Z-slabs over the listed devices (the one GPU
    of the test box named three times); files identical to the single-device run."""
    shape = (24, 28, 32)
    niftiio.write(str(tmp_path / "i.nii.gz"), synth.volume_f32(shape, 5), (0.9, 1.0, 1.1))
    niftiio.write(str(tmp_path / "m.nii.gz"), np.minimum(synth.mask_ellipsoids(shape), 1).astype(np.uint8),
                  (0.9, 1.0, 1.1))
    outs = []
    for devs in (None, "0,0,0"):
        env = dict(os.environ)
        env.pop("IFE_DEVICES", None)
        env["IFE_MULTI_TRACE"] = "1"
        if devs:
            env["IFE_DEVICES"] = devs
        base = str(tmp_path / ("o" + (devs or "single").replace(",", "")))
        r = subprocess.run([os.path.join(BIN, "ExtractFeatures"), "-i", str(tmp_path / "i.nii.gz"), "-m",
                            str(tmp_path / "m.nii.gz"), "-o", base, "-s", "1", "-s", "2.5"],
                           capture_output=True, text=True, env=env)
        assert r.returncode == 0, r.stderr
        # the tool announces its scales (SetScales): ONE upload and prepass for both of them
        # (tools/ExtractFeatures.cxx:132-154 re-executes everything per scale)
        assert r.stderr.count("ife_multi: upload and prepass") == (1 if devs else 0), r.stderr
        outs.append([niftiio.read(base + "_scale_%s%s.nii.gz" % (sc, f))[0]
                     for sc in ("1.000000", "2.500000") for f in ("GaussianBlur", "Eigenvalue1", "FrobeniusNorm")])
    for a, b in zip(*outs):
        np.testing.assert_array_equal(a, b)


@pytest.mark.gpu
def test_host_selftest(built, tmp_path):
    r = subprocess.run([os.path.join(BIN, "host_selftest"), str(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "host_selftest: ok" in r.stdout


@pytest.mark.gpu
def test_extract_features_tool(built, tmp_path, synth, oracle):
    shape, spacing = (20, 24, 28), (0.75, 0.75, 1.5)
    img = synth.volume_f32(shape, 21)
    labels = synth.mask_ellipsoids(shape)            # 0/1/2: the tool clamps to {0,1}
    niftiio.write(str(tmp_path / "img.nii.gz"), img, spacing)
    niftiio.write(str(tmp_path / "mask.nii.gz"), labels, spacing)
    r = run("ExtractFeatures", "-i", str(tmp_path / "img.nii.gz"), "-m", str(tmp_path / "mask.nii.gz"),
            "-o", str(tmp_path / "out"), "-s", "1", "--scale", "2.5")
    assert r.returncode == 0, r.stderr
    names = ["GaussianBlur", "GradientMagnitude", "Eigenvalue1", "Eigenvalue2", "Eigenvalue3",
             "LaplacianOfGaussian", "GaussianCurvature", "FrobeniusNorm"]
    mask = np.minimum(labels, 1).astype(np.uint8)
    for sig, tag in ((1.0, "1.000000"), (2.5, "2.500000")):     # std::to_string(float)
        ref = oracle.emphysema_features(img, mask, sig, spacing)
        for c, nm in enumerate(names):
            path = str(tmp_path / ("out_scale_%s%s.nii.gz" % (tag, nm)))
            assert os.path.exists(path), path
            vol, sp = niftiio.read(path)
            assert vol.dtype == np.float32 and np.allclose(sp, spacing)
            lam = np.maximum(np.abs(ref[..., 2]).astype(np.float64), 1e-30) ** (3 if c == 6 else 1)
            if c < 2:
                np.testing.assert_array_equal(vol, ref[..., c])
            else:
                assert (np.abs(vol.astype(np.float64) - ref[..., c]) / lam).max() <= 3e-6


@pytest.mark.gpu
def test_fd_and_mask_tools(built, tmp_path, synth, oracle):
    shape = (12, 16, 20)
    img = synth.volume_f32(shape, 22)
    mask = np.minimum(synth.mask_ellipsoids(shape), 1).astype(np.uint8)
    niftiio.write(str(tmp_path / "img.nii"), img)
    niftiio.write(str(tmp_path / "mask.nii"), mask)
    os.mkdir(str(tmp_path / "o"))
    env = dict(os.environ, IFE_OUT_FILE_TYPE=".nii")
    r = subprocess.run([os.path.join(BIN, "FiniteDifference_HessianFeatures"), "-i", str(tmp_path / "img.nii"),
                        "-m", str(tmp_path / "mask.nii"), "-o", str(tmp_path / "o") + "//"],
                       capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    ref = oracle.fd_hessian_features(img, mask)
    for c, nm in enumerate(["eig1", "eig2", "eig3", "LoG", "Curvature", "Frobenius"]):
        vol, _ = niftiio.read(str(tmp_path / "o" / ("hessian_%s.nii" % nm)))   # default prefix
        lam = np.maximum(np.abs(ref[..., 0]).astype(np.float64), 1e-30) ** (3 if c == 4 else 1)
        assert (np.abs(vol.astype(np.float64) - ref[..., c]) / lam).max() <= 3e-6
    r = subprocess.run([os.path.join(BIN, "FiniteDifference_GradientFeatures"), "-i", str(tmp_path / "img.nii"),
                        "-m", str(tmp_path / "mask.nii"), "-o", str(tmp_path / "o"), "-p", "g_"],
                       capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    vol, _ = niftiio.read(str(tmp_path / "o" / "g_GradientMagnitude.nii"))
    np.testing.assert_array_equal(vol, oracle.fd_gradient_features(img, mask.astype(np.float32)))
    # normalized convolution, masked output, double scale
    r = subprocess.run([os.path.join(BIN, "MaskedNormalizedConvolution"), "-i", str(tmp_path / "img.nii"),
                        "-c", str(tmp_path / "mask.nii"), "-s", "1.5", "-o", str(tmp_path / "o"),
                        "-m", "true"], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    vol, _ = niftiio.read(str(tmp_path / "o" / "normconv_scale_1.500000.nii"))
    nc = oracle.normalized_gaussian_convolution(img, mask.astype(np.float32), 1.5)
    np.testing.assert_array_equal(vol, np.where(mask != 0, nc, 0).astype(np.float32))
    # MaskedImageFilter: double pixels, outside value
    r = run("MaskedImageFilter", "-i", str(tmp_path / "img.nii"), "-m", str(tmp_path / "mask.nii"),
            "-o", str(tmp_path / "masked.nii"), "-v", "-3.5")
    assert r.returncode == 0, r.stderr
    vol, _ = niftiio.read(str(tmp_path / "masked.nii"))
    assert vol.dtype == np.float64
    np.testing.assert_array_equal(vol, np.where(mask != 0, img.astype(np.float64), -3.5))


@pytest.mark.gpu
def test_histogram_edges_tool(built, tmp_path, synth, oracle):
    """All-foreground branch against the oracle doing the tool's steps on the CPU; output
    format of tools/DetermineHistogramBinEdges_MultiScaleEigenvalueFeatures.cxx:270-296."""
    tool = "DetermineHistogramBinEdges_MultiScaleEigenvalueFeatures"
    scales, fg, nbins = [1.0, 2.5], (1, 2), 11
    cols = [[] for _ in range(16)]
    lines = []
    for k, shape in enumerate([(16, 20, 24), (12, 28, 20)]):
        img = synth.volume_f32(shape, 300 + k)
        lab = synth.mask_ellipsoids(shape)
        niftiio.write(str(tmp_path / ("img%d.nii.gz" % k)), img)
        niftiio.write(str(tmp_path / ("lab%d.nii.gz" % k)), lab)
        lines.append("%s , %s" % (tmp_path / ("img%d.nii.gz" % k), tmp_path / ("lab%d.nii.gz" % k)))
        for i, sg in enumerate(scales):
            g = oracle.gather_foreground(oracle.emphysema_features(img, np.minimum(lab, 1).astype(np.uint8), sg),
                                         lab, fg)
            for c in range(8):
                cols[i * 8 + c].append(g[c])
    (tmp_path / "pairs.csv").write_text("\n".join(lines) + "\n\n")
    out = tmp_path / "edges.txt"
    args = ["-i", str(tmp_path / "pairs.csv"), "-o", str(out), "-b", str(nbins), "-s", "1", "-s", "2.5",
            "-f", "1", "-f", "2"]
    r = run(tool, *args, "-S", "0")
    assert r.returncode == 0, r.stderr
    assert r.stdout.count("Processing") == 2
    text = out.read_text().splitlines()
    assert text[0] == ("# Features: GaussianBlur GradientMagnitude Eigenvalue1 Eigenvalue2 Eigenvalue3 "
                       "LaplacianOfGaussian GaussianCurvature FrobeniusNorm")
    assert text[1] == "# Scales: 1 2.5"
    assert len(text) == 2 + 16
    for c in range(16):
        want = oracle.equalized_edges(oracle.sort_f32(np.concatenate(cols[c])), nbins)
        assert text[2 + c] == ",".join("%g" % v for v in want), c
    # sampled branch: deterministic under IFE_SEED, edges non-decreasing and inside the range
    env = dict(os.environ, IFE_SEED="7")
    outs = []
    for rep in range(2):
        o = tmp_path / ("edges_s%d.txt" % rep)
        a = [x if x != str(out) else str(o) for x in args]
        rr = subprocess.run([os.path.join(BIN, tool)] + a + ["-S", "500"], capture_output=True, text=True, env=env)
        assert rr.returncode == 0, rr.stderr
        outs.append(o.read_text())
    assert outs[0] == outs[1]
    rows = [np.array([float(x) for x in ln.split(",")]) for ln in outs[0].splitlines()[2:]]
    assert len(rows) == 16 and all(r_.size == nbins - 1 and np.all(np.diff(r_) >= 0) for r_ in rows)
    for c in range(16):
        allv = np.concatenate(cols[c])
        assert rows[c][0] >= allv.min() - 1e-3 * abs(allv.min()) and rows[c][-1] <= allv.max() + 1e-3 * abs(allv.max())
    # more bins than samples: the reference throws std::out_of_range; here a message and EXIT_FAILURE
    bad = run(tool, *[x if x != str(nbins) else "100000" for x in args], "-S", "10")
    assert bad.returncode == 1 and "Too many bins" in bad.stderr


def test_bag_variant_flags(built):
    """The variants register exactly the flags the reference's variants read
    (tools/MakeBagDense.cxx has no -r/-R/-n, tools/MakeBagOnlyIntensity.cxx no -s)."""
    d = run("MakeBagDense", "--help").stdout
    assert "--scale" in d and "--roi-mask" in d and "--roi-file" not in d and "--num-rois" not in d
    i = run("MakeBagOnlyIntensity", "--help").stdout
    assert "--roi-file" in i and "--num-rois" in i and "--scale" not in i
    assert run("MakeBagDense", "-i", "a", "-m", "b", "-H", "c", "-o", "d", "-s", "1", "-n", "3").returncode == 1
    assert run("MakeBagOnlyIntensity", "-i", "a", "-m", "b", "-H", "c", "-o", "d", "-s", "1").returncode == 1


@pytest.mark.gpu
def test_makebag_dense_and_only_intensity(built, tmp_path, synth, oracle):
    """MakeBagDense: one region per mask voxel whose box fits, in raster order
    (DenseROIGenerator.hxx:24-46), rows against the oracle.  MakeBagOnlyIntensity: the image
    itself binned by one histogram (MakeBagOnlyIntensity.cxx:355-389)."""
    shape, scales, nbins = (10, 12, 14), [1.0], 5
    img = synth.volume_f32(shape, 402)
    lab = np.zeros(shape, np.uint16)
    lab[3:7, 2:9, 4:11] = 2
    lab[0, 0, 0] = 1                                  # a mask voxel whose box does not fit
    niftiio.write(str(tmp_path / "img.nii.gz"), img)
    niftiio.write(str(tmp_path / "lab.nii.gz"), lab)
    clamped = np.minimum(lab, 1).astype(np.uint8)
    feat = oracle.emphysema_features(img, clamped, 1.0)
    edges = np.stack([oracle.equalized_edges(oracle.sort_f32(feat[..., c][clamped != 0]), nbins)
                      for c in range(8)])
    edges32 = np.array([[np.float32(float("%.9g" % v)) for v in row] for row in edges], np.float32)
    (tmp_path / "hist.txt").write_text("".join(",".join("%.9g" % v for v in row) + "\n" for row in edges))
    os.mkdir(str(tmp_path / "out"))
    sx, sy, sz = 5, 3, 3
    r = run("MakeBagDense", "-i", str(tmp_path / "img.nii.gz"), "-m", str(tmp_path / "lab.nii.gz"),
            "-H", str(tmp_path / "hist.txt"), "-o", str(tmp_path / "out"), "-s", "1",
            "-x", str(sx), "-y", str(sy), "-z", str(sz), "-p", "dense")
    assert r.returncode == 0, r.stderr
    want_boxes = []
    for z, y, x in zip(*np.nonzero(lab)):             # np.nonzero walks z, y, x: raster order
        x0, y0, z0 = x - sx // 2, y - sy // 2, z - sz // 2
        if x0 >= 0 and y0 >= 0 and z0 >= 0 and x0 + sx <= 14 and y0 + sy <= 12 and z0 + sz <= 10:
            want_boxes.append((x0, y0, z0, sx, sy, sz))
    info = (tmp_path / "out" / "dense.ROIInfo").read_text().splitlines()
    assert info == ["[%d, %d, %d][%d, %d, %d]" % b for b in want_boxes] and len(info) == 4 * 7 * 7
    rows = (tmp_path / "out" / "dense.bag").read_text().splitlines()
    assert len(rows) == len(want_boxes)
    boxes = np.array(want_boxes, np.int64)
    for j in range(0, len(rows), 17):
        _, fr = oracle.roi_histograms(feat, clamped, boxes[j:j + 1], edges32)
        assert [t.replace("-nan", "nan") for t in rows[j].split(",")] == ["%g" % v for v in fr.ravel()], j
    # intensity only: one histogram, regions from a file
    iedges = np.array([-500.0, 0.0, 500.0, 1500.0], np.float32)
    (tmp_path / "ihist.txt").write_text("# intensity\n" + ",".join("%.9g" % v for v in iedges) + "\n")
    (tmp_path / "rois.txt").write_text("index size\n[1, 1, 1][6, 5, 4]\n[7, 6, 5][7, 6, 5]\n[0, 0, 7][3, 3, 3]\n")
    r = run("MakeBagOnlyIntensity", "-i", str(tmp_path / "img.nii.gz"), "-m", str(tmp_path / "lab.nii.gz"),
            "-H", str(tmp_path / "ihist.txt"), "-o", str(tmp_path / "out"), "-r", str(tmp_path / "rois.txt"),
            "-p", "int")
    assert r.returncode == 0, r.stderr
    rows = (tmp_path / "out" / "int.bag").read_text().splitlines()
    assert len(rows) == 3
    for row, (x0, y0, z0, bx, by, bz) in zip(rows, ((1, 1, 1, 6, 5, 4), (7, 6, 5, 7, 6, 5), (0, 0, 7, 3, 3, 3))):
        v = img[z0:z0 + bz, y0:y0 + by, x0:x0 + bx][lab[z0:z0 + bz, y0:y0 + by, x0:x0 + bx] != 0]
        cnt = np.bincount(np.searchsorted(iedges, v, side="left"), minlength=5)  # (-inf,e0], (e0,e1], ...
        fr = cnt.astype(np.float32) / np.float32(cnt.sum()) if cnt.sum() else np.full(5, np.nan, np.float32)
        assert [t.replace("-nan", "nan") for t in row.split(",")] == ["%g" % f for f in fr]
    two = run("MakeBagOnlyIntensity", "-i", str(tmp_path / "img.nii.gz"), "-m", str(tmp_path / "lab.nii.gz"),
              "-H", str(tmp_path / "hist.txt"), "-o", str(tmp_path / "out"), "-r", str(tmp_path / "rois.txt"))
    assert two.returncode == 1 and "Expected exactly one histogram" in two.stderr


def test_makebag_usage(built):
    h = run("MakeBag", "--help")
    assert h.returncode == 0
    assert all(f in h.stdout for f in ("--histogram-spec", "--outdir", "--roi-file", "--roi-file-has-header",
                                       "--roi-mask", "--roi-mask-value", "--num-rois", "--roi-size-x", "--prefix"))
    assert run("MakeBag", "-i", "x").returncode == 1


@pytest.mark.gpu
def test_makebag_tool(built, tmp_path, synth, oracle):
    """ROI file branch against the oracle (bag rows of tools/MakeBag.cxx:405-486), then the
    generated-ROI branch: .ROIInfo in the reader's own format, boxes inside, centred on mask."""
    shape, scales, nbins = (24, 28, 32), [1.0, 2.0], 7
    img = synth.volume_f32(shape, 401)
    lab = synth.mask_ellipsoids(shape).astype(np.uint16)
    niftiio.write(str(tmp_path / "img.nii.gz"), img)
    niftiio.write(str(tmp_path / "lab.nii.gz"), lab)
    clamped = np.minimum(lab, 1).astype(np.uint8)
    feats = [oracle.emphysema_features(img, clamped, s) for s in scales]
    edges = np.stack([oracle.equalized_edges(oracle.sort_f32(f[..., c][clamped != 0]), nbins)
                      for f in feats for c in range(8)])
    spec = "# Features: ...\n# Scales: 1 2\n" + "".join(",".join("%.9g" % v for v in row) + "\n" for row in edges)
    (tmp_path / "hist.txt").write_text(spec)
    rng = np.random.default_rng(5)
    rois = np.stack([rng.integers(0, 32 - 9, 6), rng.integers(0, 28 - 8, 6), rng.integers(0, 24 - 7, 6)], 1)
    (tmp_path / "rois.txt").write_text("index size\n" + "".join("[%d, %d, %d][9, 8, 7]\n" % tuple(r) for r in rois))
    os.mkdir(str(tmp_path / "out"))
    r = run("MakeBag", "-i", str(tmp_path / "img.nii.gz"), "-m", str(tmp_path / "lab.nii.gz"),
            "-H", str(tmp_path / "hist.txt"), "-o", str(tmp_path / "out"), "-s", "1", "-s", "2",
            "-r", str(tmp_path / "rois.txt"), "-p", "case1")
    assert r.returncode == 0, r.stderr
    assert "Got 6 rois." in r.stdout and r.stdout.count("Skipping a line") == 2
    boxes = np.concatenate([rois, np.tile([9, 8, 7], (6, 1))], 1)
    rows = (tmp_path / "out" / "case1.bag").read_text().splitlines()
    assert len(rows) == 6
    edges32 = np.array([[np.float32(float("%.9g" % v)) for v in row] for row in edges], np.float32)
    for j, row in enumerate(rows):
        want = []
        for i, f in enumerate(feats):
            _, fr = oracle.roi_histograms(f, clamped, boxes[j:j + 1], edges32[i * 8:(i + 1) * 8])
            want += ["%g" % v for v in fr.ravel()]
        # a region without mask voxels gives 0/0: x86 prints the default NaN as "-nan"
        assert [t.replace("-nan", "nan") for t in row.split(",")] == want, j
    # generated regions
    env = dict(os.environ, IFE_SEED="11")
    r = subprocess.run([os.path.join(BIN, "MakeBag"), "-i", str(tmp_path / "img.nii.gz"), "-m",
                        str(tmp_path / "lab.nii.gz"), "-H", str(tmp_path / "hist.txt"), "-o", str(tmp_path / "out"),
                        "-s", "1", "-s", "2", "-n", "5", "-x", "7", "-y", "5", "-z", "3", "-p", "gen"],
                       capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    info = (tmp_path / "out" / "gen.ROIInfo").read_text().splitlines()
    assert len(info) == 5 and len((tmp_path / "out" / "gen.bag").read_text().splitlines()) == 5
    import re
    for ln in info:
        m = re.fullmatch(r"\[(\d+), (\d+), (\d+)\]\[7, 5, 3\]", ln)
        assert m, ln
        x, y, z = (int(v) for v in m.groups())
        assert x + 7 <= 32 and y + 5 <= 28 and z + 3 <= 24
        assert lab[z + 1, y + 2, x + 3] != 0          # centre = start + size/2 lies on the mask
    # wrong number of histogram rows
    r = run("MakeBag", "-i", str(tmp_path / "img.nii.gz"), "-m", str(tmp_path / "lab.nii.gz"),
            "-H", str(tmp_path / "hist.txt"), "-o", str(tmp_path / "out"), "-s", "1", "-r", str(tmp_path / "rois.txt"))
    assert r.returncode == 1 and "Number of histograms must match" in r.stderr
