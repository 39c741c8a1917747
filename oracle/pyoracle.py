"""ctypes binding of the CPU oracle (oracle/libife_oracle.so).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg; the product package never imports this module.
Function-by-function citations of the reference live in ``ife_oracle.c``.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libife_oracle.so")

TRIG_CMATH = 0
TRIG_MATH_H = 1
DSCALE_ITK = 0
DSCALE_POW = 1


class Dims(C.Structure):
    _fields_ = [("nx", C.c_int64), ("ny", C.c_int64), ("nz", C.c_int64),
                ("sx", C.c_double), ("sy", C.c_double), ("sz", C.c_double)]


class GaussCoeffs(C.Structure):
    _fields_ = [(n, C.c_double) for n in (
        "N0", "N1", "N2", "N3", "D1", "D2", "D3", "D4", "M1", "M2", "M3", "M4",
        "BN1", "BN2", "BN3", "BN4", "BM1", "BM2", "BM3", "BM4")]


def build():
    """Compile the oracle with gcc (oracle/Makefile)."""
    subprocess.check_call(["make", "-s", "-C", _HERE])


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.ife_or_get_threads.restype = C.c_int
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _dims(shape_zyx, spacing_xyz=(1.0, 1.0, 1.0)):
    nz, ny, nx = shape_zyx
    return Dims(nx, ny, nz, *[float(s) for s in spacing_xyz])


def _chk(rc, what):
    if rc != 0:
        raise RuntimeError("oracle %s failed rc=%d" % (what, rc))


def set_threads(n):
    lib().ife_or_set_threads(int(n))


def eig3(A6, trig_mode=TRIG_CMATH):
    """A6: (..., 6) float32 or float64 -> (..., 3) eigenvalues, |.| descending."""
    A6 = np.ascontiguousarray(A6)
    n = A6.size // 6
    if A6.dtype == np.float64:
        out = np.empty(A6.shape[:-1] + (3,), np.float64)
        lib().ife_or_eig3_batch_f64(_p(A6, C.c_double), C.c_int64(n), _p(out, C.c_double))
    else:
        A6 = np.ascontiguousarray(A6, np.float32)
        out = np.empty(A6.shape[:-1] + (3,), np.float32)
        lib().ife_or_eig3_batch_f32(_p(A6, C.c_float), C.c_int64(n), _p(out, C.c_float),
                                    C.c_int(trig_mode))
    return out


def eigfeat(A6, trig_mode=TRIG_CMATH):
    A6 = np.ascontiguousarray(A6, np.float32)
    n = A6.size // 6
    out = np.empty(A6.shape[:-1] + (6,), np.float32)
    lib().ife_or_eigfeat_batch_f32(_p(A6, C.c_float), C.c_int64(n), _p(out, C.c_float),
                                   C.c_int(trig_mode))
    return out


def gauss_coeffs(sigma, spacing=1.0):
    c = GaussCoeffs()
    _chk(lib().ife_or_gauss_coeffs_zero_order(C.c_double(sigma), C.c_double(spacing), C.byref(c)),
         "gauss_coeffs")
    return c


def gauss_coeffs_order(sigma, spacing=1.0, order=0):
    """Orders 0, 1, 2 of itk::RecursiveGaussianImageFilter::SetUp (row f4)."""
    c = GaussCoeffs()
    _chk(lib().ife_or_gauss_coeffs_order(C.c_double(sigma), C.c_double(spacing), C.c_int(order),
                                   C.byref(c)), "gauss_coeffs")
    return c


def iir_line_order(data, sigma, spacing=1.0, order=0):
    data = np.ascontiguousarray(data, np.float64)
    c = gauss_coeffs_order(sigma, spacing, order)
    out = np.empty_like(data)
    scr = np.empty_like(data)
    lib().ife_or_iir_line(_p(data, C.c_double), _p(out, C.c_double), _p(scr, C.c_double),
                          C.c_int64(data.size), C.byref(c))
    return out


def differential_normalized_convolution(image, certainty, sigma, axis_xyz, spacing=(1.0, 1.0, 1.0)):
    image = np.ascontiguousarray(image, np.float32)
    certainty = np.ascontiguousarray(certainty, np.float32)
    out = np.empty_like(image)
    d = _dims(image.shape, spacing)
    _chk(lib().ife_or_differential_normalized_convolution(
        _p(image, C.c_float), _p(certainty, C.c_float), _p(out, C.c_float), C.byref(d),
        C.c_double(sigma), C.c_int(axis_xyz)), "differential normconv")
    return out


def iir_line(data, sigma, spacing=1.0):
    data = np.ascontiguousarray(data, np.float64)
    c = gauss_coeffs(sigma, spacing)
    out = np.empty_like(data)
    scr = np.empty_like(data)
    lib().ife_or_iir_line(_p(data, C.c_double), _p(out, C.c_double), _p(scr, C.c_double),
                          C.c_int64(data.size), C.byref(c))
    return out


def recursive_gaussian_axis(vol, axis_xyz, sigma, spacing=(1.0, 1.0, 1.0)):
    vol = np.ascontiguousarray(vol, np.float32)
    out = np.empty_like(vol)
    d = _dims(vol.shape, spacing)
    _chk(lib().ife_or_recursive_gaussian_axis(_p(vol, C.c_float), _p(out, C.c_float), C.byref(d),
                                              C.c_int(axis_xyz), C.c_double(sigma)), "iir axis")
    return out


def smoothing_recursive_gaussian(vol, sigma, spacing=(1.0, 1.0, 1.0)):
    vol = np.ascontiguousarray(vol, np.float32)
    out = np.empty_like(vol)
    d = _dims(vol.shape, spacing)
    _chk(lib().ife_or_smoothing_recursive_gaussian(_p(vol, C.c_float), _p(out, C.c_float),
                                                   C.byref(d), C.c_double(sigma)), "smoothing")
    return out


def normalized_gaussian_convolution(image, certainty, sigma, spacing=(1.0, 1.0, 1.0)):
    image = np.ascontiguousarray(image, np.float32)
    certainty = np.ascontiguousarray(certainty, np.float32)
    out = np.empty_like(image)
    d = _dims(image.shape, spacing)
    _chk(lib().ife_or_normalized_gaussian_convolution(
        _p(image, C.c_float), _p(certainty, C.c_float), _p(out, C.c_float), C.byref(d),
        C.c_double(sigma)), "normconv")
    return out


def derivative(vol, order, direction_xyz, spacing=(1.0, 1.0, 1.0), dscale=DSCALE_ITK):
    vol = np.ascontiguousarray(vol, np.float32)
    out = np.empty_like(vol)
    d = _dims(vol.shape, spacing)
    _chk(lib().ife_or_derivative(_p(vol, C.c_float), _p(out, C.c_float), C.byref(d),
                                 C.c_int(order), C.c_int(direction_xyz), C.c_int(dscale)),
         "derivative")
    return out


def hessian3d(vol, spacing=(1.0, 1.0, 1.0), dscale=DSCALE_ITK):
    vol = np.ascontiguousarray(vol, np.float32)
    out = np.empty(vol.shape + (6,), np.float32)
    d = _dims(vol.shape, spacing)
    _chk(lib().ife_or_hessian3d(_p(vol, C.c_float), _p(out, C.c_float), C.byref(d),
                                C.c_int(dscale)), "hessian3d")
    return out


def gradient_magnitude(vol, spacing=(1.0, 1.0, 1.0)):
    vol = np.ascontiguousarray(vol, np.float32)
    out = np.empty_like(vol)
    d = _dims(vol.shape, spacing)
    _chk(lib().ife_or_gradient_magnitude(_p(vol, C.c_float), _p(out, C.c_float), C.byref(d)),
         "gradmag")
    return out


def emphysema_features(image, mask, sigma, spacing=(1.0, 1.0, 1.0), trig_mode=TRIG_CMATH,
                       dscale=DSCALE_ITK):
    image = np.ascontiguousarray(image, np.float32)
    mask = np.ascontiguousarray(mask, np.uint8)
    out = np.empty(image.shape + (8,), np.float32)
    d = _dims(image.shape, spacing)
    _chk(lib().ife_or_emphysema_features(
        _p(image, C.c_float), _p(mask, C.c_uint8), _p(out, C.c_float), C.byref(d),
        C.c_double(sigma), C.c_int(trig_mode), C.c_int(dscale)), "emphysema_features")
    return out


def fd_hessian_features(image, mask, spacing=(1.0, 1.0, 1.0), trig_mode=TRIG_CMATH,
                        dscale=DSCALE_ITK):
    image = np.ascontiguousarray(image, np.float32)
    out = np.empty(image.shape + (6,), np.float32)
    d = _dims(image.shape, spacing)
    if mask is None:
        mp = None
    else:
        mask = np.ascontiguousarray(mask, np.uint8)
        mp = _p(mask, C.c_uint8)
    _chk(lib().ife_or_fd_hessian_features(_p(image, C.c_float), mp, _p(out, C.c_float),
                                          C.byref(d), C.c_int(trig_mode), C.c_int(dscale)),
         "fd_hessian_features")
    return out


def fd_gradient_features(image, mask_f32, spacing=(1.0, 1.0, 1.0)):
    image = np.ascontiguousarray(image, np.float32)
    mask_f32 = np.ascontiguousarray(mask_f32, np.float32)
    out = np.empty_like(image)
    d = _dims(image.shape, spacing)
    _chk(lib().ife_or_fd_gradient_features(_p(image, C.c_float), _p(mask_f32, C.c_float),
                                           _p(out, C.c_float), C.byref(d)), "fd_gradient")
    return out


def mask_image_f64(image, mask, outside=0.0):
    image = np.ascontiguousarray(image, np.float64)
    mask = np.ascontiguousarray(mask, np.float64)
    out = np.empty_like(image)
    lib().ife_or_mask_image_f64(_p(image, C.c_double), _p(mask, C.c_double), C.c_double(outside),
                                _p(out, C.c_double), C.c_int64(image.size))
    return out


# ---- rows f1 / f2: histogram edges and dense histograms ----

class EdgeWalkError(RuntimeError):
    """rc 1: fewer samples than bins (std::out_of_range in the reference); rc 3: the walk
    would run past the last sample (an assert in the reference)."""

    def __init__(self, rc):
        RuntimeError.__init__(self, "equalized edges rc=%d" % rc)
        self.rc = rc


def equalized_edges(sorted_values, nbins):
    """determineEdgesForEqualizedHistogram on an ascending float32/float64 array."""
    v = np.ascontiguousarray(sorted_values)
    if v.dtype == np.float64:
        out = np.empty(max(nbins - 1, 0), np.float64)
        rc = lib().ife_or_equalized_edges_f64(_p(v, C.c_double), C.c_int64(v.size),
                                              C.c_int64(nbins), _p(out, C.c_double))
    else:
        v = np.ascontiguousarray(v, np.float32)
        out = np.empty(max(nbins - 1, 0), np.float32)
        rc = lib().ife_or_equalized_edges_f32(_p(v, C.c_float), C.c_int64(v.size),
                                              C.c_int64(nbins), _p(out, C.c_float))
    if rc != 0:
        raise EdgeWalkError(rc)
    return out


def sort_f32(values):
    v = np.array(values, np.float32).ravel()
    lib().ife_or_sort_f32(_p(v, C.c_float), C.c_int64(v.size))
    return v


def gather_foreground(features, mask, foreground):
    """features (..., ncomp) interleaved float32, mask uint8 -> (ncomp, m) sample columns."""
    features = np.ascontiguousarray(features, np.float32)
    mask = np.ascontiguousarray(mask, np.uint8)
    ncomp = features.shape[-1]
    fg = np.ascontiguousarray(foreground, np.uint32)
    f = lib().ife_or_gather_foreground
    f.restype = C.c_int64
    m = f(_p(features, C.c_float), C.c_int(ncomp), _p(mask, C.c_uint8), C.c_int64(mask.size),
          _p(fg, C.c_uint32), C.c_int(fg.size), None)
    cols = np.empty((ncomp, m), np.float32)
    ptrs = (C.POINTER(C.c_float) * ncomp)(*[_p(cols[c], C.c_float) for c in range(ncomp)])
    f(_p(features, C.c_float), C.c_int(ncomp), _p(mask, C.c_uint8), C.c_int64(mask.size),
      _p(fg, C.c_uint32), C.c_int(fg.size), ptrs)
    return cols


def dense_histogram(edges, values):
    edges = np.ascontiguousarray(edges, np.float32)
    values = np.ascontiguousarray(values, np.float32).ravel()
    counts = np.empty(edges.size + 1, np.uint32)
    freqs = np.empty(edges.size + 1, np.float32)
    lib().ife_or_dense_histogram_f32(_p(edges, C.c_float), C.c_int64(edges.size),
                                     _p(values, C.c_float), C.c_int64(values.size),
                                     _p(counts, C.c_uint32), _p(freqs, C.c_float))
    return counts, freqs


def roi_histograms(features, mask, rois, edges):
    """features (nz, ny, nx, ncomp), mask uint8, rois (n, 6) x,y,z,sx,sy,sz, edges (ncomp, ne)
    -> counts (n, ncomp, ne+1) uint32, freqs float32: the bag rows of MakeBag for one scale."""
    features = np.ascontiguousarray(features, np.float32)
    mask = np.ascontiguousarray(mask, np.uint8)
    rois = np.ascontiguousarray(rois, np.int64).reshape(-1, 6)
    edges = np.ascontiguousarray(edges, np.float32)
    ncomp, ne = edges.shape
    d = _dims(mask.shape)
    counts = np.empty((rois.shape[0], ncomp, ne + 1), np.uint32)
    freqs = np.empty((rois.shape[0], ncomp, ne + 1), np.float32)
    rc = lib().ife_or_roi_histograms(_p(features, C.c_float), C.c_int(ncomp), _p(mask, C.c_uint8),
                                     C.byref(d), _p(rois, C.c_int64), C.c_int(rois.shape[0]),
                                     _p(edges, C.c_float), C.c_int64(ne), _p(counts, C.c_uint32),
                                     _p(freqs, C.c_float))
    _chk(rc, "roi_histograms")
    return counts, freqs


# ---- oracle/_ref: the reference's own statistics / IO headers, compiled in place ----

_REF_PATH = os.path.join(_HERE, "_ref", "libife_ref_stats.so")
_ref = None


def build_ref():
    """`make -C oracle _ref`: builds only where /root/reference exists."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "_ref"])


def ref_lib():
    """The compiled reference (or None where neither it nor /root/reference is present)."""
    global _ref
    if _ref is None:
        if not os.path.exists(_REF_PATH):
            build_ref()
        if not os.path.exists(_REF_PATH):
            return None
        _ref = C.CDLL(_REF_PATH)
    return _ref


def ref_equalized_edges(sorted_values, nbins):
    """Call the reference template itself.  Only for inputs on which equalized_edges()
    succeeds: the reference asserts (aborts) where that returns rc 3."""
    v = np.ascontiguousarray(sorted_values)
    r = ref_lib()
    if v.dtype == np.float64:
        out = np.empty(max(nbins - 1, 0), np.float64)
        rc = r.ife_ref_edges_f64(_p(v, C.c_double), C.c_size_t(v.size), C.c_size_t(nbins),
                                 _p(out, C.c_double))
    else:
        v = np.ascontiguousarray(v, np.float32)
        out = np.empty(max(nbins - 1, 0), np.float32)
        rc = r.ife_ref_edges_f32(_p(v, C.c_float), C.c_size_t(v.size), C.c_size_t(nbins),
                                 _p(out, C.c_float))
    if rc != 0:
        raise EdgeWalkError(rc)
    return out


def ref_dense_histogram(edges, values):
    edges = np.ascontiguousarray(edges, np.float32)
    values = np.ascontiguousarray(values, np.float32).ravel()
    counts = np.empty(edges.size + 1, np.uint32)
    freqs = np.empty(edges.size + 1, np.float32)
    ref_lib().ife_ref_dense_histogram_f32(_p(edges, C.c_float), C.c_size_t(edges.size),
                                          _p(values, C.c_float), C.c_size_t(values.size),
                                          _p(counts, C.c_uint32), _p(freqs, C.c_float))
    return counts, freqs


def ref_write_sequence(values, sep=","):
    v = np.ascontiguousarray(values, np.float32).ravel()
    buf = C.create_string_buffer(32 * v.size + 16)
    n = ref_lib().ife_ref_write_sequence_f32(_p(v, C.c_float), C.c_size_t(v.size),
                                             C.c_char(sep.encode()), buf, C.c_size_t(len(buf)))
    if n < 0:
        raise RuntimeError("buffer too small")
    return buf.value.decode()


def ref_read_pair_list(path, sep=","):
    buf = C.create_string_buffer(1 << 16)
    n = ref_lib().ife_ref_read_pair_list(path.encode(), C.c_char(sep.encode()), buf,
                                         C.c_size_t(len(buf)))
    if n == -2:
        raise ValueError("Line does not contain a separator")
    if n < 0:
        raise RuntimeError("buffer too small")
    return [tuple(line.split("\t")) for line in buf.value.decode().splitlines()]
