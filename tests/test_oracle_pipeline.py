"""CPU tests of the oracle's ITK-wired stages (a3..a8).

PARITY UNPINNED: the reference holds no test or fixture for these stages and ITK is not
available, so they are checked against analytic identities that any correct restatement
of the published algorithms must satisfy, and against self-pinned snapshots
(tests/golden/make_golden.py) that guard against accidental change.
"""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
FLT_MAX = np.finfo(np.float32).max


# ---- recursive Gaussian ---------------------------------------------------------------
@pytest.mark.parametrize("sd", [0.5, 1.0, 2.0, 4.0, 8.0])
def test_impulse_response_properties(oracle, sd):
    n = 513
    x = np.zeros(n)
    x[n // 2] = 1.0
    h = oracle.iir_line(x, sd)
    assert abs(h.sum() - 1.0) < 1e-9                      # unit DC gain
    assert np.abs(h - h[::-1]).max() < 1e-15              # symmetric
    g = np.exp(-0.5 * ((np.arange(n) - n // 2) / sd) ** 2) / (sd * np.sqrt(2 * np.pi))
    assert np.abs(h - g).max() < {0.5: 1e-2, 1.0: 1.3e-3, 2.0: 6e-4, 4.0: 3e-4, 8.0: 1.5e-4}[sd]


@pytest.mark.parametrize("n", [4, 5, 7, 16, 33])
@pytest.mark.parametrize("sd", [0.6, 1.0, 3.0, 10.0])
def test_constant_line_is_preserved_to_rounding(oracle, n, sd):
    """Edge extension: a constant line comes back constant (to double rounding), and an
    all-ones line rounds to exactly 1.0f -- the fact the NULL-mask shortcut relies on."""
    out = oracle.iir_line(np.full(n, 3.25), sd)
    assert np.abs(out - 3.25).max() < 1e-12
    ones = oracle.iir_line(np.ones(n), sd)
    assert (ones.astype(np.float32) == np.float32(1.0)).all()


def test_all_ones_certainty_smooths_to_exact_one(oracle):
    for shape, sigma, sp in [((9, 10, 11), 1.0, (1, 1, 1)), ((8, 8, 40), 4.0, (0.7, 0.7, 1.3)),
                             ((4, 4, 4), 0.6, (1, 1, 1))]:
        g = oracle.smoothing_recursive_gaussian(np.ones(shape, np.float32), sigma, sp)
        assert (g == np.float32(1.0)).all()


def test_impulse_fixture(oracle):
    fx = json.load(open(os.path.join(HERE, "golden", "iir_impulse.json")))["responses"]
    for sd, hexes in fx.items():
        x = np.zeros(65)
        x[32] = 1.0
        got = oracle.iir_line(x, float(sd))
        assert [float.hex(float(v)) for v in got] == hexes


def test_smoothing_is_z_then_x_then_y_with_float_between(oracle, synth):
    shape = (9, 12, 10)
    v = synth.volume_f32(shape, 1)
    sp = (0.8, 1.0, 1.25)
    a = oracle.recursive_gaussian_axis(v, 2, 1.7, sp)
    a = oracle.recursive_gaussian_axis(a, 0, 1.7, sp)
    a = oracle.recursive_gaussian_axis(a, 1, 1.7, sp)
    np.testing.assert_array_equal(oracle.smoothing_recursive_gaussian(v, 1.7, sp), a)
    # the axis order matters at float resolution (that is why it is part of the contract)
    b = oracle.recursive_gaussian_axis(v, 0, 1.7, sp)
    b = oracle.recursive_gaussian_axis(b, 1, 1.7, sp)
    b = oracle.recursive_gaussian_axis(b, 2, 1.7, sp)
    assert np.abs(a - b).max() < 1e-2 and (a != b).any()


def test_axis_pass_matches_line_filter(oracle, synth):
    shape = (6, 7, 8)
    v = synth.volume_f32(shape, 2)
    out = oracle.recursive_gaussian_axis(v, 1, 2.0, (1, 0.5, 1))   # along y, sigma_d = 4
    line = oracle.iir_line(v[3, :, 5].astype(np.float64), 2.0, 0.5).astype(np.float32)
    np.testing.assert_array_equal(out[3, :, 5], line)


def test_short_axis_rejected(oracle):
    with pytest.raises(RuntimeError):
        oracle.smoothing_recursive_gaussian(np.zeros((3, 8, 8), np.float32), 1.0)


# ---- normalized convolution ---------------------------------------------------------------
def test_normalized_convolution_definition(oracle, synth):
    shape = (10, 11, 12)
    img = synth.volume_f32(shape, 3)
    c = (synth.mask_ellipsoids(shape) > 0).astype(np.float32)
    c[0, 0, 0] = 1
    out = oracle.normalized_gaussian_convolution(img, c, 1.5)
    num = oracle.smoothing_recursive_gaussian(img * c, 1.5)
    den = oracle.smoothing_recursive_gaussian(c, 1.5)
    exp = np.where(den != 0, num / np.where(den != 0, den, 1), FLT_MAX).astype(np.float32)
    np.testing.assert_array_equal(out, exp)
    # certainty one everywhere: plain smoothing
    ones = np.ones(shape, np.float32)
    np.testing.assert_array_equal(oracle.normalized_gaussian_convolution(img, ones, 1.5),
                                  oracle.smoothing_recursive_gaussian(img, 1.5))
    # zero certainty: ITK's Div functor returns max()
    assert (oracle.normalized_gaussian_convolution(img, np.zeros(shape, np.float32), 1.5)
            == FLT_MAX).all()


# ---- derivatives ------------------------------------------------------------------------------
def _poly(shape, sp, co):
    z, y, x = np.meshgrid(*[np.arange(n, dtype=np.float64) for n in shape], indexing="ij")
    x, y, z = x * sp[0], y * sp[1], z * sp[2]
    a, b, c, d, e, f = co
    return (a * x * x + b * y * y + c * z * z + d * x * y + e * x * z + f * y * z).astype(np.float32)


def test_hessian_of_quadratic_is_exact_in_the_interior(oracle):
    shape = (9, 10, 11)
    co = (1.0, -2.0, 0.5, 3.0, -1.0, 4.0)  # integer-valued samples: exact in float
    H = oracle.hessian3d(_poly(shape, (1, 1, 1), co))
    inner = H[1:-1, 1:-1, 1:-1]
    a, b, c, d, e, f = co
    for k, v in enumerate([2 * a, d, e, 2 * b, f, 2 * c]):   # xx, xy, xz, yy, yz, zz
        assert (inner[..., k] == np.float32(v)).all()


def test_derivative_spacing_modes(oracle):
    shape = (7, 8, 9)
    sp = (0.5, 2.0, 4.0)
    v = _poly(shape, sp, (1, 1, 1, 1, 1, 1))
    # first derivative of x^2 + ... at interior: 2x + y + z (central difference exact)
    dx = oracle.derivative(v, 1, 0, sp)
    z, y, x = np.meshgrid(*[np.arange(n, dtype=np.float64) for n in shape], indexing="ij")
    exp = 2 * x * sp[0] + y * sp[1] + z * sp[2]
    np.testing.assert_allclose(dx[:, :, 1:-1], exp[:, :, 1:-1], rtol=1e-6)
    # second derivative: ITK scales the operator by 1/spacing once (mode 0)
    dxx_itk = oracle.derivative(v, 2, 0, sp, oracle.DSCALE_ITK)
    dxx_pow = oracle.derivative(v, 2, 0, sp, oracle.DSCALE_POW)
    np.testing.assert_allclose(dxx_pow[:, :, 1:-1], 2.0, rtol=1e-5)
    np.testing.assert_allclose(dxx_itk[:, :, 1:-1], 2.0 * sp[0], rtol=1e-5)


def test_replicate_boundary(oracle):
    v = np.arange(5, dtype=np.float32)[None, None, :] * np.ones((3, 3, 1), np.float32)
    dx = oracle.derivative(v, 1, 0)
    assert dx[1, 1].tolist() == [0.5, 1.0, 1.0, 1.0, 0.5]
    dxx = oracle.derivative(v, 2, 0)
    assert dxx[1, 1].tolist() == [1.0, 0.0, 0.0, 0.0, -1.0]
    # chained cross term clamps at each stage
    w = (np.arange(4, dtype=np.float32)[None, :, None] * np.arange(5, dtype=np.float32)[None, None, :]
         * np.ones((2, 1, 1), np.float32))
    dxy = oracle.hessian3d(w)[..., 1]
    assert dxy[0, 1, 2] == 1.0 and dxy[0, 0, 0] == 0.25 and dxy[0, 0, 2] == 0.5


def test_gradient_magnitude_of_ramp(oracle):
    shape = (6, 7, 8)
    sp = (0.5, 1.0, 2.0)
    z, y, x = np.meshgrid(*[np.arange(n, dtype=np.float64) for n in shape], indexing="ij")
    v = (3 * x * sp[0] + 4 * y * sp[1] + 12 * z * sp[2]).astype(np.float32)
    g = oracle.gradient_magnitude(v, sp)
    assert (g[1:-1, 1:-1, 1:-1] == 13.0).all()


# ---- composite filter ---------------------------------------------------------------------------
def test_constant_image_gives_constant_blur_and_zero_features(oracle):
    shape = (8, 9, 10)
    img = np.full(shape, 7.5, np.float32)
    mask = np.ones(shape, np.uint8)
    out = oracle.emphysema_features(img, mask, 2.0)
    assert (out[..., 0] == 7.5).all() and (out[..., 1:] == 0).all()


def test_mask_zeroes_outside_and_weights_inside(oracle, synth):
    shape = (12, 13, 14)
    img = synth.volume_f32(shape, 4)
    labels = synth.mask_ellipsoids(shape)
    labels[0, 0, 0] = 2
    out = oracle.emphysema_features(img, labels, 1.0)
    assert (out[labels == 0] == 0).all()
    # label 2 weighs twice as much as label 1 in the normalized convolution
    clamped = oracle.emphysema_features(img, np.minimum(labels, 1).astype(np.uint8), 1.0)
    assert (out != clamped).any()


def test_power_of_two_scaling_is_exact(oracle, synth):
    """Every rounding commutes with a power-of-two scale: features scale by 2, the
    eigenvalue product by 8."""
    shape = (10, 10, 10)
    img = synth.volume_f32(shape, 5)
    mask = np.ones(shape, np.uint8)
    a = oracle.emphysema_features(img, mask, 1.5)
    b = oracle.emphysema_features(img * np.float32(2), mask, 1.5)
    for k in (0, 1, 2, 3, 4, 5, 7):
        np.testing.assert_array_equal(b[..., k], a[..., k] * np.float32(2))
    np.testing.assert_array_equal(b[..., 6], a[..., 6] * np.float32(8))


def test_fd_tools(oracle, synth):
    shape = (8, 9, 10)
    img = synth.volume_f32(shape, 6)
    mask = np.minimum(synth.mask_ellipsoids(shape), 1).astype(np.uint8)
    f = oracle.fd_hessian_features(img, mask)
    H = oracle.hessian3d(img)
    np.testing.assert_array_equal(f[mask != 0], oracle.eigfeat(H)[mask != 0])
    assert (f[mask == 0] == 0).all()
    g = oracle.fd_gradient_features(img, mask.astype(np.float32))
    np.testing.assert_array_equal(g, np.where(mask != 0, oracle.gradient_magnitude(img), 0))
    d = oracle.mask_image_f64(img.astype(np.float64), mask.astype(np.float64), -1.0)
    np.testing.assert_array_equal(d, np.where(mask != 0, img.astype(np.float64), -1.0))


# ---- self-pinned snapshots ------------------------------------------------------------------------
def test_pipeline_snapshot(oracle):
    z = np.load(os.path.join(HERE, "golden", "pipeline_24x20x28.npz"))
    for s, sigma in enumerate(z["sigmas"]):
        got = oracle.emphysema_features(z["image"], z["mask"], float(sigma), tuple(z["spacing"]))
        np.testing.assert_array_equal(got, z["features"][s])


def test_fd_snapshot(oracle):
    z = np.load(os.path.join(HERE, "golden", "fd_hessian_20.npz"))
    np.testing.assert_array_equal(oracle.fd_hessian_features(z["image"], z["mask"]), z["features"])
    np.testing.assert_array_equal(oracle.hessian3d(z["image"]), z["hessian"])
    np.testing.assert_array_equal(oracle.gradient_magnitude(z["image"]), z["gradmag"])


# ---------------------------------------------------------------------------------
# Row f4: first / second order recursive Gaussian and the differential normalized convolution
# (no reference implementation exists: NormalizedGaussianConvolutionImageFilter.h:28-44 is a
# comment; these pin the restated ITK coefficient sets by the properties ITK normalises them to)
# ---------------------------------------------------------------------------------
@pytest.mark.parametrize("sigma", [1.0, 2.0, 4.0])
def test_derivative_filters_have_their_defining_responses(oracle, sigma):
    x = np.arange(300, dtype=np.float64)
    mid = slice(100, 200)
    d1 = oracle.iir_line_order(3.0 * x + 5.0, sigma, 1.0, 1)
    assert np.abs(d1[mid] - 3.0).max() < 1e-10            # unit response to a unit ramp
    d2 = oracle.iir_line_order(0.5 * x * x, sigma, 1.0, 2)
    assert np.abs(d2[mid] - 1.0).max() < 1e-8             # unit response to a unit parabola
    assert np.abs(oracle.iir_line_order(np.full(300, 7.0), sigma, 1.0, 1)).max() < 1e-12
    assert np.abs(oracle.iir_line_order(np.full(300, 7.0), sigma, 1.0, 2)[mid]).max() < 1e-10
    # impulse responses: antisymmetric / symmetric, close to the sampled Gaussian derivatives
    imp = np.zeros(301)
    imp[150] = 1.0
    k = np.arange(-150, 151, dtype=np.float64)
    g = np.exp(-k * k / (2 * sigma * sigma)) / (sigma * np.sqrt(2 * np.pi))
    h1 = oracle.iir_line_order(imp, sigma, 1.0, 1)
    h2 = oracle.iir_line_order(imp, sigma, 1.0, 2)
    assert np.abs(h1 + h1[::-1]).max() < 1e-12 and np.abs(h2 - h2[::-1]).max() < 1e-12
    if sigma >= 2.0:
        assert np.abs(h1 - (-k / sigma ** 2) * g).max() < 0.005 * np.abs(k / sigma ** 2 * g).max()
        assert np.abs(h2 - (k * k / sigma ** 4 - 1 / sigma ** 2) * g).max() < 0.02 / sigma ** 3
    # spacing: sigma in physical units, response still per PIXEL (ITK's gradient filter divides later)
    d1s = oracle.iir_line_order(3.0 * x, 2.0 * sigma, 2.0, 1)
    assert np.abs(d1s[mid] - 3.0).max() < 1e-10


def test_differential_normalized_convolution_properties(oracle):
    rng = np.random.default_rng(4)
    shape = (20, 24, 28)
    z, y, x = np.meshgrid(*[np.arange(n, dtype=np.float32) for n in shape], indexing="ij")
    ramp = (2.0 * x - 3.0 * y + 0.5 * z).astype(np.float32)
    ones = np.ones(shape, np.float32)
    sp = (0.5, 2.0, 1.0)
    inner = (slice(8, 12), slice(9, 15), slice(10, 18))
    for axis, slope in ((0, 2.0 / 0.5), (1, -3.0 / 2.0), (2, 0.5 / 1.0)):   # physical units
        d = oracle.differential_normalized_convolution(ramp, ones, 1.5, axis, sp)
        # (the volume is only a few sigma wide: the replicated border bends the ramp a little)
        assert np.abs(d[inner] - slope).max() < 1e-2 * abs(slope)
    # full certainty: equals the first-order recursive Gaussian of the image itself / spacing
    img = rng.standard_normal(shape).astype(np.float32)
    d = oracle.differential_normalized_convolution(img, ones, 2.0, 0)
    smooth = oracle.differential_normalized_convolution(img, ones, 2.0, 0)
    np.testing.assert_array_equal(d, smooth)
    # a scaled certainty cancels; a hole in the certainty changes the result only nearby
    d_half = oracle.differential_normalized_convolution(img, 0.5 * ones, 2.0, 0)
    assert np.abs(d_half - d).max() < 1e-4
    holed = ones.copy()
    holed[2:4, 2:4, 2:4] = 0
    d_h = oracle.differential_normalized_convolution(img, holed, 1.0, 1)
    d_f = oracle.differential_normalized_convolution(img, ones, 1.0, 1)
    assert np.abs(d_h - d_f)[12:, 14:, 16:].max() < 1e-5 and np.abs(d_h - d_f).max() > 1e-3
    assert (oracle.differential_normalized_convolution(img, 0 * ones, 1.0, 2) == np.finfo(np.float32).max).all()
