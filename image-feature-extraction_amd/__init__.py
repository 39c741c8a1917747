"""image-feature-extraction_amd -- MI355X (gfx950) per-voxel Hessian feature engine.

Thin ctypes binding of the C-ABI in ``include/ife_hip.h`` (``csrc/libife_hip.so``).
The product is the HIP library and the C++ host mirror under ``host/``; this module
exists so that tests and ``bench.py`` can call the same entry points from Python.

There is no CPU path here: importing works without a GPU (so that the symbol table
can be checked), but creating a :class:`Context` raises when the library or a gfx950
device is missing.

The package directory name contains a hyphen (it mirrors the reference's repository
name); import it with ``importlib.import_module("image-feature-extraction_amd")``.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# IFE_HIP_LIB selects another build of the same library (kernel tuning experiments)
LIB_PATH = os.environ.get("IFE_HIP_LIB") or os.path.join(_HERE, "csrc", "libife_hip.so")

# enums of include/ife_hip.h
OK, E_ARG, E_SIZE, E_HIP, E_NOMEM, E_STATE = 0, -1, -2, -3, -4, -5
F32, I16, U8, U16 = 0, 1, 2, 3
INTERLEAVED, PLANAR = 0, 1
MEM_HOST, MEM_DEVICE = 0, 1
OPT_TRIG_MODE, OPT_DSCALE_MODE, OPT_PROFILE, OPT_ZCHUNK, OPT_IIR_BLOCK, OPT_IIR_CKPT, OPT_IIR_FMA = 1, 2, 3, 4, 5, 6, 7
OPT_FUSED_DIVIDE = 8
OPT_CONST_LINES = 9
OPT_FEAT_RING = 10
Z_STATE_BYTES = 32   # IFE_Z_STATE_BYTES: one state record of the slab Z pass, per line and job
Z_OVERLAP_LO, Z_OVERLAP_HI = 3, 4  # IFE_Z_OVERLAP_*: neighbour planes around a slab's input
NUM_FEATURES = 8
FEATURE_NAMES = ("GaussianBlur", "GradientMagnitude", "Eigenvalue1", "Eigenvalue2",
                 "Eigenvalue3", "LaplacianOfGaussian", "GaussianCurvature", "FrobeniusNorm")

# every symbol include/ife_hip.h declares
EXPORTS = (
    "ife_abi_version", "ife_ctx_create", "ife_ctx_destroy", "ife_last_error",
    "ife_ctx_set_stream", "ife_ctx_set_option", "ife_ctx_reserve", "ife_ctx_synchronize",
    "ife_eigenvalues", "ife_eigenvalue_features", "ife_hessian3d", "ife_gradient_magnitude",
    "ife_normalized_gaussian_convolution", "ife_differential_normalized_convolution",
    "ife_emphysema_features",
    "ife_emphysema_features_begin", "ife_emphysema_features_fetch", "ife_emphysema_features_end",
    "ife_fd_hessian_features", "ife_fd_gradient_features", "ife_mask_image_f64",
    "ife_get_kernel_times", "ife_reset_kernel_times", "ife_measure_stream",
    "ife_stage_prepare", "ife_stage_recursive_gaussian", "ife_stage_recursive_gaussian_batch",
    "ife_stage_features", "ife_stage_z_ck_bytes", "ife_stage_z_sweep", "ife_stage_z_combine",
    "ife_stage_z_fused", "ife_stage_recursive_gaussian_quotient",
    "ife_multi_create", "ife_multi_destroy", "ife_multi_last_error", "ife_multi_set_option",
    "ife_multi_emphysema_features", "ife_multi_emphysema_features_begin",
    "ife_multi_emphysema_features_fetch", "ife_multi_emphysema_features_end",
    "ife_sort_f32", "ife_equalized_edges_f32", "ife_equalized_edges_f64", "ife_dense_histogram_f32", "ife_roi_histograms", "ife_bag_image",
    "ife_samples_create", "ife_samples_destroy", "ife_samples_count", "ife_samples_clear",
    "ife_samples_add_features", "ife_samples_add_image", "ife_samples_sort",
    "ife_samples_equalized_edges", "ife_samples_read_column",
)


class VolumeDesc(C.Structure):
    _fields_ = [("nx", C.c_int64), ("ny", C.c_int64), ("nz", C.c_int64),
                ("sx", C.c_double), ("sy", C.c_double), ("sz", C.c_double)]


class KernelTime(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_int64), ("total_ms", C.c_double)]


class IfeError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("ife error %d: %s" % (code, msg))
        self.code = code


_lib = None


def load_library():
    """dlopen csrc/libife_hip.so.  Fails loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s is missing: build it with `make -C %s` (hipcc --offload-arch=gfx950); "
            "there is no CPU fallback" % (LIB_PATH, os.path.dirname(LIB_PATH)))
    # A torch wheel carries its own HIP runtime under the same SONAME as the system one this
    # library links to, and whichever copy is loaded first serves the whole process.  Ours
    # first leaves torch on a runtime it was not built for ("No HIP GPUs are available" at its
    # first CUDA call), so where torch is installed it is imported before the library.
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64, f32p = C.c_void_p, C.c_int, C.c_int64, C.c_void_p
    vd = C.POINTER(VolumeDesc)
    lib.ife_abi_version.restype = i32
    lib.ife_ctx_create.argtypes = [i32, C.POINTER(vp)]
    lib.ife_ctx_destroy.argtypes = [vp]
    lib.ife_ctx_destroy.restype = None
    lib.ife_last_error.argtypes = [vp]
    lib.ife_last_error.restype = C.c_char_p
    lib.ife_ctx_set_stream.argtypes = [vp, vp]
    lib.ife_ctx_set_option.argtypes = [vp, i32, i32]
    lib.ife_ctx_reserve.argtypes = [vp, vd]
    lib.ife_ctx_synchronize.argtypes = [vp]
    lib.ife_eigenvalues.argtypes = [vp, f32p, i64, f32p, i32]
    lib.ife_eigenvalue_features.argtypes = [vp, f32p, i64, f32p, i32]
    lib.ife_hessian3d.argtypes = [vp, f32p, vd, f32p, i32, i32]
    lib.ife_gradient_magnitude.argtypes = [vp, f32p, vd, f32p, i32]
    lib.ife_normalized_gaussian_convolution.argtypes = [vp, f32p, f32p, vd, C.c_double, f32p, i32]
    lib.ife_differential_normalized_convolution.argtypes = [vp, f32p, f32p, vd, C.c_double, i32, f32p, i32]
    lib.ife_emphysema_features.argtypes = [vp, vp, i32, vp, i32, vd, C.POINTER(C.c_float), i32,
                                           f32p, i32, i32]
    lib.ife_emphysema_features_begin.argtypes = [vp, vp, i32, vp, i32, vd, C.POINTER(C.c_float), i32, i32]
    lib.ife_emphysema_features_fetch.argtypes = [vp, i32, f32p]
    lib.ife_emphysema_features_end.argtypes = [vp]
    lib.ife_fd_hessian_features.argtypes = [vp, vp, i32, vp, i32, vd, f32p, i32, i32]
    lib.ife_fd_gradient_features.argtypes = [vp, f32p, f32p, vd, f32p, i32]
    lib.ife_mask_image_f64.argtypes = [vp, vp, vp, C.c_double, i64, vp, i32]
    lib.ife_stage_prepare.argtypes = [vp, vp, i32, vp, i32, vd, i32, vp, vp]
    lib.ife_stage_recursive_gaussian.argtypes = [vp, vp, vp, vd, i32, C.c_double]
    lib.ife_stage_recursive_gaussian_batch.argtypes = [vp, i32, C.POINTER(vp), C.POINTER(vp), vd,
                                                       i32, C.POINTER(C.c_double), i32]
    lib.ife_stage_features.argtypes = [vp, vp, vp, vp, i32, vd, i32, i32, vp, i32]
    lib.ife_stage_z_ck_bytes.argtypes = [vd]
    lib.ife_stage_z_ck_bytes.restype = C.c_size_t
    lib.ife_stage_z_sweep.argtypes = [vp, i32, i32, C.POINTER(vp), vd, i64, i64,
                                      C.POINTER(C.c_double), i32, vp, vp, C.POINTER(vp)]
    lib.ife_stage_z_combine.argtypes = [vp, i32, C.POINTER(vp), C.POINTER(vp), vd, i64, i64,
                                        C.POINTER(C.c_double), i32, i32, C.POINTER(vp)]
    lib.ife_stage_z_fused.argtypes = [vp, i32, i32, C.POINTER(vp), C.POINTER(vp), vd, i64, i64,
                                      C.POINTER(C.c_double), i32, i32, vp, vp, C.POINTER(vp)]
    lib.ife_stage_recursive_gaussian_quotient.argtypes = [vp, i32, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp),
                                                          vd, i32, C.POINTER(C.c_double)]
    lib.ife_multi_create.argtypes = [C.POINTER(i32), i32, C.POINTER(vp)]
    lib.ife_multi_destroy.argtypes = [vp]
    lib.ife_multi_destroy.restype = None
    lib.ife_multi_last_error.argtypes = [vp]
    lib.ife_multi_last_error.restype = C.c_char_p
    lib.ife_multi_set_option.argtypes = [vp, i32, i32]
    lib.ife_multi_emphysema_features.argtypes = [vp, vp, i32, vp, i32, vd, C.POINTER(C.c_float), i32, f32p, i32]
    lib.ife_multi_emphysema_features_begin.argtypes = [vp, vp, i32, vp, i32, vd, C.POINTER(C.c_float), i32, i32]
    lib.ife_multi_emphysema_features_fetch.argtypes = [vp, i32, f32p]
    lib.ife_multi_emphysema_features_end.argtypes = [vp]
    lib.ife_get_kernel_times.argtypes = [vp, C.POINTER(KernelTime), i32]
    lib.ife_reset_kernel_times.argtypes = [vp]
    lib.ife_measure_stream.argtypes = [vp, i32, vp, vp, C.c_size_t, i32, C.POINTER(C.c_double)]
    lib.ife_sort_f32.argtypes = [vp, vp, i64, vp, i32]
    lib.ife_equalized_edges_f32.argtypes = [vp, vp, i64, i32, vp, i32]
    lib.ife_equalized_edges_f64.argtypes = [vp, vp, i64, i32, vp, i32]
    lib.ife_dense_histogram_f32.argtypes = [vp, vp, i32, vp, i64, vp, i32]
    lib.ife_roi_histograms.argtypes = [vp, vp, i32, i32, vp, i32, vd, vp, i32, vp, i32, vp, i32]
    lib.ife_bag_image.argtypes = [vp, vp, i32, vp, i32, vd, C.POINTER(C.c_float), i32, vp, i32, vp,
                                  i32, vp, i32]
    lib.ife_samples_create.argtypes = [vp, i32, C.POINTER(vp)]
    lib.ife_samples_destroy.argtypes = [vp]
    lib.ife_samples_destroy.restype = None
    lib.ife_samples_count.argtypes = [vp, i32, C.POINTER(i64)]
    lib.ife_samples_clear.argtypes = [vp]
    lib.ife_samples_add_features.argtypes = [vp, vp, i32, vp, i32, i32, vp, i32, i64, vp, i32,
                                             vp, i64, i32]
    lib.ife_samples_add_image.argtypes = [vp, vp, vp, i32, vp, i32, vd, C.POINTER(C.c_float),
                                          i32, vp, i32, vp, i64, i32]
    lib.ife_samples_sort.argtypes = [vp, vp]
    lib.ife_samples_equalized_edges.argtypes = [vp, vp, i32, vp]
    lib.ife_samples_read_column.argtypes = [vp, vp, i32, vp, i64]
    _lib = lib
    return lib


def _desc(shape_zyx, spacing_xyz):
    nz, ny, nx = (int(v) for v in shape_zyx)
    sx, sy, sz = (float(v) for v in spacing_xyz)
    return VolumeDesc(nx, ny, nz, sx, sy, sz)


_IMG_DT = {np.dtype(np.float32): F32, np.dtype(np.int16): I16}
_MSK_DT = {np.dtype(np.uint8): U8, np.dtype(np.uint16): U16}


class Context:
    """One ``ife_ctx``: a device, a stream, a workspace.  Single-owner."""

    def __init__(self, device=0):
        self._lib = load_library()
        h = C.c_void_p()
        rc = self._lib.ife_ctx_create(int(device), C.byref(h))
        if rc != OK:
            raise IfeError(rc, self._lib.ife_last_error(None).decode())
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._lib.ife_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _chk(self, rc):
        if rc < 0:
            raise IfeError(rc, self._lib.ife_last_error(self._h).decode())
        return rc

    # ---- configuration -----------------------------------------------------
    def set_stream(self, stream_handle):
        self._chk(self._lib.ife_ctx_set_stream(self._h, C.c_void_p(stream_handle or 0)))

    def set_option(self, option, value):
        self._chk(self._lib.ife_ctx_set_option(self._h, int(option), int(value)))

    def reserve(self, shape_zyx, spacing_xyz=(1.0, 1.0, 1.0)):
        d = _desc(shape_zyx, spacing_xyz)
        self._chk(self._lib.ife_ctx_reserve(self._h, C.byref(d)))

    def synchronize(self):
        self._chk(self._lib.ife_ctx_synchronize(self._h))

    def kernel_times(self):
        arr = (KernelTime * 16)()
        n = self._chk(self._lib.ife_get_kernel_times(self._h, arr, 16))
        return {arr[i].name.decode(): (int(arr[i].launches), float(arr[i].total_ms))
                for i in range(n)}

    def measure_stream(self, mode, dst_ptr, src_ptr, nbytes, reps=5):
        """GB/s of a 16-byte-per-lane fill (mode 0: bytes written) or copy (mode 1: bytes read +
        written) over device buffers (ife_measure_stream)."""
        ms = C.c_double()
        self._chk(self._lib.ife_measure_stream(self._h, int(mode), C.c_void_p(dst_ptr),
                                               C.c_void_p(src_ptr or 0), C.c_size_t(nbytes),
                                               int(reps), C.byref(ms)))
        return (1 if mode == 0 else 2) * nbytes / (ms.value * 1e-3) / 1e9

    def reset_kernel_times(self):
        self._chk(self._lib.ife_reset_kernel_times(self._h))

    # ---- host-array entry points (numpy in, numpy out) -----------------------
    def eigenvalues(self, A6):
        A6 = np.ascontiguousarray(A6, np.float32)
        n = A6.size // 6
        out = np.empty(A6.shape[:-1] + (3,), np.float32)
        self._chk(self._lib.ife_eigenvalues(self._h, A6.ctypes.data, n, out.ctypes.data, MEM_HOST))
        return out

    def eigenvalue_features(self, A6):
        A6 = np.ascontiguousarray(A6, np.float32)
        n = A6.size // 6
        out = np.empty(A6.shape[:-1] + (6,), np.float32)
        self._chk(self._lib.ife_eigenvalue_features(self._h, A6.ctypes.data, n, out.ctypes.data,
                                                    MEM_HOST))
        return out

    def hessian3d(self, image, spacing=(1.0, 1.0, 1.0), layout=INTERLEAVED):
        image = np.ascontiguousarray(image, np.float32)
        d = _desc(image.shape, spacing)
        out = np.empty(image.shape + (6,) if layout == INTERLEAVED else (6,) + image.shape,
                       np.float32)
        self._chk(self._lib.ife_hessian3d(self._h, image.ctypes.data, C.byref(d), out.ctypes.data,
                                          layout, MEM_HOST))
        return out

    def gradient_magnitude(self, image, spacing=(1.0, 1.0, 1.0)):
        image = np.ascontiguousarray(image, np.float32)
        d = _desc(image.shape, spacing)
        out = np.empty_like(image)
        self._chk(self._lib.ife_gradient_magnitude(self._h, image.ctypes.data, C.byref(d),
                                                   out.ctypes.data, MEM_HOST))
        return out

    def normalized_gaussian_convolution(self, image, certainty, sigma, spacing=(1.0, 1.0, 1.0)):
        image = np.ascontiguousarray(image, np.float32)
        certainty = np.ascontiguousarray(certainty, np.float32)
        d = _desc(image.shape, spacing)
        out = np.empty_like(image)
        self._chk(self._lib.ife_normalized_gaussian_convolution(
            self._h, image.ctypes.data, certainty.ctypes.data, C.byref(d), float(sigma),
            out.ctypes.data, MEM_HOST))
        return out

    def differential_normalized_convolution(self, image, certainty, sigma, axis_xyz,
                                            spacing=(1.0, 1.0, 1.0)):
        image = np.ascontiguousarray(image, np.float32)
        certainty = np.ascontiguousarray(certainty, np.float32)
        d = _desc(image.shape, spacing)
        out = np.empty_like(image)
        self._chk(self._lib.ife_differential_normalized_convolution(
            self._h, image.ctypes.data, certainty.ctypes.data, C.byref(d), float(sigma),
            int(axis_xyz), out.ctypes.data, MEM_HOST))
        return out

    def emphysema_features(self, image, mask, sigmas, spacing=(1.0, 1.0, 1.0),
                           layout=INTERLEAVED):
        """One 8-component volume per sigma: shape (S, nz, ny, nx, 8) or (S, 8, nz, ny, nx)."""
        image = np.ascontiguousarray(image)
        if image.dtype not in _IMG_DT:
            image = image.astype(np.float32)
        mdt, mptr = U8, None
        if mask is not None:
            mask = np.ascontiguousarray(mask)
            if mask.dtype not in _MSK_DT:
                raise TypeError("mask must be uint8 or uint16")
            mdt, mptr = _MSK_DT[mask.dtype], mask.ctypes.data
        d = _desc(image.shape, spacing)
        sig = (C.c_float * len(sigmas))(*[float(s) for s in sigmas])
        shp = image.shape + (8,) if layout == INTERLEAVED else (8,) + image.shape
        out = np.empty((len(sigmas),) + shp, np.float32)
        self._chk(self._lib.ife_emphysema_features(
            self._h, image.ctypes.data, _IMG_DT[image.dtype], mptr, mdt, C.byref(d), sig,
            len(sigmas), out.ctypes.data, layout, MEM_HOST))
        return out

    def emphysema_features_stream(self, image, mask, sigmas, spacing=(1.0, 1.0, 1.0),
                                  layout=INTERLEAVED):
        """Generator over the scales: begin once, yield one 8-component volume per sigma."""
        image = np.ascontiguousarray(image)
        if image.dtype not in _IMG_DT:
            image = image.astype(np.float32)
        mdt, mptr = U8, None
        if mask is not None:
            mask = np.ascontiguousarray(mask)
            mdt, mptr = _MSK_DT[mask.dtype], mask.ctypes.data
        d = _desc(image.shape, spacing)
        sig = (C.c_float * len(sigmas))(*[float(s) for s in sigmas])
        self._chk(self._lib.ife_emphysema_features_begin(
            self._h, image.ctypes.data, _IMG_DT[image.dtype], mptr, mdt, C.byref(d), sig,
            len(sigmas), layout))
        try:
            shp = image.shape + (8,) if layout == INTERLEAVED else (8,) + image.shape
            for k in range(len(sigmas)):
                out = np.empty(shp, np.float32)
                self._chk(self._lib.ife_emphysema_features_fetch(self._h, k, out.ctypes.data))
                yield out
        finally:
            self._chk(self._lib.ife_emphysema_features_end(self._h))

    def fd_hessian_features(self, image, mask=None, spacing=(1.0, 1.0, 1.0), layout=INTERLEAVED):
        image = np.ascontiguousarray(image)
        if image.dtype not in _IMG_DT:
            image = image.astype(np.float32)
        mdt, mptr = U8, None
        if mask is not None:
            mask = np.ascontiguousarray(mask)
            if mask.dtype not in _MSK_DT:
                raise TypeError("mask must be uint8 or uint16")
            mdt, mptr = _MSK_DT[mask.dtype], mask.ctypes.data
        d = _desc(image.shape, spacing)
        out = np.empty(image.shape + (6,) if layout == INTERLEAVED else (6,) + image.shape,
                       np.float32)
        self._chk(self._lib.ife_fd_hessian_features(
            self._h, image.ctypes.data, _IMG_DT[image.dtype], mptr, mdt, C.byref(d),
            out.ctypes.data, layout, MEM_HOST))
        return out

    def fd_gradient_features(self, image, mask_f32=None, spacing=(1.0, 1.0, 1.0)):
        image = np.ascontiguousarray(image, np.float32)
        mptr = None
        if mask_f32 is not None:
            mask_f32 = np.ascontiguousarray(mask_f32, np.float32)
            mptr = mask_f32.ctypes.data
        d = _desc(image.shape, spacing)
        out = np.empty_like(image)
        self._chk(self._lib.ife_fd_gradient_features(self._h, image.ctypes.data, mptr, C.byref(d),
                                                     out.ctypes.data, MEM_HOST))
        return out

    def mask_image_f64(self, image, mask, outside=0.0):
        image = np.ascontiguousarray(image, np.float64)
        mask = np.ascontiguousarray(mask, np.float64)
        out = np.empty_like(image)
        self._chk(self._lib.ife_mask_image_f64(self._h, image.ctypes.data, mask.ctypes.data,
                                               float(outside), image.size, out.ctypes.data,
                                               MEM_HOST))
        return out

    # ---- device-pointer entry points (inputs already in HBM) ------------------
    def emphysema_features_device(self, image_ptr, image_dtype, mask_ptr, mask_dtype, shape_zyx,
                                  spacing, sigmas, out_ptr, layout=INTERLEAVED):
        d = _desc(shape_zyx, spacing)
        sig = (C.c_float * len(sigmas))(*[float(s) for s in sigmas])
        self._chk(self._lib.ife_emphysema_features(
            self._h, C.c_void_p(image_ptr), image_dtype, C.c_void_p(mask_ptr or 0), mask_dtype,
            C.byref(d), sig, len(sigmas), C.c_void_p(out_ptr), layout, MEM_DEVICE))

    def fd_hessian_features_device(self, image_ptr, image_dtype, mask_ptr, mask_dtype, shape_zyx,
                                   spacing, out_ptr, layout=INTERLEAVED):
        d = _desc(shape_zyx, spacing)
        self._chk(self._lib.ife_fd_hessian_features(
            self._h, C.c_void_p(image_ptr), image_dtype, C.c_void_p(mask_ptr or 0), mask_dtype,
            C.byref(d), C.c_void_p(out_ptr), layout, MEM_DEVICE))

    # ---- stage entry points (device pointers; Z-slab orchestration, slab.py) ------------
    def stage_prepare(self, image_ptr, image_dtype, mask_ptr, mask_dtype, slab_shape_zyx, tc_ptr,
                      cf_ptr, y_chunks=1):
        d = _desc(slab_shape_zyx, (1.0, 1.0, 1.0))
        self._chk(self._lib.ife_stage_prepare(
            self._h, C.c_void_p(image_ptr), image_dtype, C.c_void_p(mask_ptr or 0), mask_dtype,
            C.byref(d), int(y_chunks), C.c_void_p(tc_ptr), C.c_void_p(cf_ptr or 0)))

    def stage_recursive_gaussian(self, in_ptr, out_ptr, shape_zyx, spacing, axis_xyz, sigma):
        d = _desc(shape_zyx, spacing)
        self._chk(self._lib.ife_stage_recursive_gaussian(
            self._h, C.c_void_p(in_ptr), C.c_void_p(out_ptr), C.byref(d), int(axis_xyz),
            float(sigma)))

    def stage_recursive_gaussian_batch(self, in_ptrs, out_ptrs, shape_zyx, spacing, axis_xyz,
                                       sigmas, in_y_chunks=1):
        """One launch over len(in_ptrs) float volumes of the same shape (<= 8 jobs)."""
        n = len(in_ptrs)
        d = _desc(shape_zyx, spacing)
        ins = (C.c_void_p * n)(*in_ptrs)
        outs = (C.c_void_p * n)(*out_ptrs)
        sg = (C.c_double * n)(*[float(s) for s in sigmas])
        self._chk(self._lib.ife_stage_recursive_gaussian_batch(
            self._h, n, ins, outs, C.byref(d), int(axis_xyz), sg, int(in_y_chunks)))

    def stage_features(self, num_ptr, den_ptr, mask_ptr, mask_dtype, slab_shape_zyx, spacing,
                       halo_lo, halo_hi, out_ptr, layout=INTERLEAVED):
        d = _desc(slab_shape_zyx, spacing)
        self._chk(self._lib.ife_stage_features(
            self._h, C.c_void_p(num_ptr), C.c_void_p(den_ptr or 0), C.c_void_p(mask_ptr or 0),
            mask_dtype, C.byref(d), int(bool(halo_lo)), int(bool(halo_hi)), C.c_void_p(out_ptr),
            layout))

    def stage_z_ck_bytes(self, slab_shape_zyx):
        d = _desc(slab_shape_zyx, (1.0, 1.0, 1.0))
        return int(self._lib.ife_stage_z_ck_bytes(C.byref(d)))

    def stage_z_sweep(self, direction, in_ptrs, slab_shape_zyx, spacing, line0, nlines, sigmas,
                      has_neighbour, state_in_ptr, state_out_ptr, ck_ptrs):
        """Causal (direction 0) or anticausal (1) sweep of the Z pass of a Z-slab."""
        n = len(in_ptrs)
        d = _desc(slab_shape_zyx, spacing)
        ins = (C.c_void_p * n)(*in_ptrs)
        cks = (C.c_void_p * n)(*ck_ptrs)
        sg = (C.c_double * n)(*[float(s) for s in sigmas])
        self._chk(self._lib.ife_stage_z_sweep(
            self._h, int(direction), n, ins, C.byref(d), int(line0), int(nlines), sg,
            int(bool(has_neighbour)), C.c_void_p(state_in_ptr or 0), C.c_void_p(state_out_ptr), cks))

    def stage_z_combine(self, in_ptrs, out_ptrs, slab_shape_zyx, spacing, line0, nlines, sigmas,
                        has_lo, has_hi, ck_ptrs):
        n = len(in_ptrs)
        d = _desc(slab_shape_zyx, spacing)
        ins = (C.c_void_p * n)(*in_ptrs)
        outs = (C.c_void_p * n)(*out_ptrs)
        cks = (C.c_void_p * n)(*ck_ptrs)
        sg = (C.c_double * n)(*[float(s) for s in sigmas])
        self._chk(self._lib.ife_stage_z_combine(
            self._h, n, ins, outs, C.byref(d), int(line0), int(nlines), sg, int(bool(has_lo)),
            int(bool(has_hi)), cks))

    def stage_z_fused(self, direction, in_ptrs, out_ptrs, slab_shape_zyx, spacing, line0, nlines,
                      sigmas, has_lo, has_hi, state_in_ptr, state_out_ptr, ck_ptrs):
        """Sweep of `direction` (0 causal, 1 anticausal) and combine in one launch."""
        n = len(in_ptrs)
        d = _desc(slab_shape_zyx, spacing)
        ins = (C.c_void_p * n)(*in_ptrs)
        outs = (C.c_void_p * n)(*out_ptrs)
        cks = (C.c_void_p * n)(*ck_ptrs)
        sg = (C.c_double * n)(*[float(s) for s in sigmas])
        self._chk(self._lib.ife_stage_z_fused(
            self._h, int(direction), n, ins, outs, C.byref(d), int(line0), int(nlines), sg,
            int(bool(has_lo)), int(bool(has_hi)), C.c_void_p(state_in_ptr or 0),
            C.c_void_p(state_out_ptr), cks))

    def stage_recursive_gaussian_quotient(self, num_ptrs, den_ptrs, out_ptrs, shape_zyx, spacing,
                                          axis_xyz, sigmas):
        """Last axis pass in its quotient form: out[j] = G(num[j]) / G(den[j]) (<= 4 jobs)."""
        n = len(num_ptrs)
        d = _desc(shape_zyx, spacing)
        nums = (C.c_void_p * n)(*num_ptrs)
        dens = (C.c_void_p * n)(*den_ptrs)
        outs = (C.c_void_p * n)(*out_ptrs)
        sg = (C.c_double * n)(*[float(s) for s in sigmas])
        self._chk(self._lib.ife_stage_recursive_gaussian_quotient(
            self._h, n, nums, dens, outs, C.byref(d), int(axis_xyz), sg))

    # ---- rows f1 / f2: sample columns, histogram edges, dense histograms --------------
    def sort_f32(self, values):
        v = np.ascontiguousarray(values, np.float32).ravel()
        out = np.empty_like(v)
        self._chk(self._lib.ife_sort_f32(self._h, v.ctypes.data, v.size, out.ctypes.data, MEM_HOST))
        return out

    def sort_f32_device(self, in_ptr, n, out_ptr):
        self._chk(self._lib.ife_sort_f32(self._h, C.c_void_p(in_ptr), int(n), C.c_void_p(out_ptr),
                                         MEM_DEVICE))

    def equalized_edges(self, sorted_values, nbins):
        """determineEdgesForEqualizedHistogram on an ascending float32 (or float64) array."""
        v = np.ascontiguousarray(sorted_values)
        if v.dtype == np.float64:
            out = np.empty(max(int(nbins) - 1, 0), np.float64)
            self._chk(self._lib.ife_equalized_edges_f64(self._h, v.ctypes.data, v.size, int(nbins),
                                                        out.ctypes.data, MEM_HOST))
            return out
        v = np.ascontiguousarray(v, np.float32).ravel()
        out = np.empty(max(int(nbins) - 1, 0), np.float32)
        self._chk(self._lib.ife_equalized_edges_f32(self._h, v.ctypes.data, v.size, int(nbins),
                                                    out.ctypes.data, MEM_HOST))
        return out

    def dense_histogram(self, edges, values):
        e = np.ascontiguousarray(edges, np.float32).ravel()
        v = np.ascontiguousarray(values, np.float32).ravel()
        counts = np.empty(e.size + 1, np.uint32)
        self._chk(self._lib.ife_dense_histogram_f32(self._h, e.ctypes.data, e.size, v.ctypes.data,
                                                    v.size, counts.ctypes.data, MEM_HOST))
        return counts

    def roi_histograms(self, features, mask, rois, edges, spacing=(1.0, 1.0, 1.0), layout=INTERLEAVED):
        """Bag rows of one feature volume: counts (n_rois, ncomp, n_edges+1) uint32."""
        f = np.ascontiguousarray(features, np.float32)
        mask = np.ascontiguousarray(mask)
        rois = np.ascontiguousarray(rois, np.int64).reshape(-1, 6)
        edges = np.ascontiguousarray(edges, np.float32)
        ncomp, ne = edges.shape
        d = _desc(mask.shape, spacing)
        counts = np.empty((rois.shape[0], ncomp, ne + 1), np.uint32)
        self._chk(self._lib.ife_roi_histograms(
            self._h, f.ctypes.data, layout, ncomp, mask.ctypes.data, _MSK_DT[mask.dtype], C.byref(d),
            rois.ctypes.data, rois.shape[0], edges.ctypes.data, ne, counts.ctypes.data, MEM_HOST))
        return counts

    def bag_image(self, image, mask, sigmas, rois, edges, spacing=(1.0, 1.0, 1.0)):
        """One image of MakeBag: counts (n_rois, n_sigmas*8, n_edges+1) uint32."""
        image = np.ascontiguousarray(image)
        if image.dtype not in _IMG_DT:
            image = image.astype(np.float32)
        mask = np.ascontiguousarray(mask)
        rois = np.ascontiguousarray(rois, np.int64).reshape(-1, 6)
        edges = np.ascontiguousarray(edges, np.float32)
        ne = edges.shape[1]
        d = _desc(image.shape, spacing)
        sig = (C.c_float * len(sigmas))(*[float(s) for s in sigmas])
        counts = np.empty((rois.shape[0], 8 * len(sigmas), ne + 1), np.uint32)
        self._chk(self._lib.ife_bag_image(
            self._h, image.ctypes.data, _IMG_DT[image.dtype], mask.ctypes.data, _MSK_DT[mask.dtype],
            C.byref(d), sig, len(sigmas), rois.ctypes.data, rois.shape[0], edges.ctypes.data, ne,
            counts.ctypes.data, MEM_HOST))
        return counts

    def samples(self, n_columns):
        return Samples(self, n_columns)


class Multi:
    """``ife_multi``: the Z-slab engine over several devices of one process (peer copies)."""

    def __init__(self, devices):
        self._lib = load_library()
        devs = (C.c_int * len(devices))(*[int(d) for d in devices])
        h = C.c_void_p()
        rc = self._lib.ife_multi_create(devs, len(devices), C.byref(h))
        if rc != OK:
            raise IfeError(rc, self._lib.ife_multi_last_error(None).decode())
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._lib.ife_multi_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _chk(self, rc):
        if rc < 0:
            raise IfeError(rc, self._lib.ife_multi_last_error(self._h).decode())

    def set_option(self, option, value):
        self._chk(self._lib.ife_multi_set_option(self._h, int(option), int(value)))

    def emphysema_features(self, image, mask, sigmas, spacing=(1.0, 1.0, 1.0), layout=INTERLEAVED):
        image = np.ascontiguousarray(image)
        if image.dtype not in _IMG_DT:
            image = image.astype(np.float32)
        mdt, mptr = U8, None
        if mask is not None:
            mask = np.ascontiguousarray(mask)
            mdt, mptr = _MSK_DT[mask.dtype], mask.ctypes.data
        d = _desc(image.shape, spacing)
        sig = (C.c_float * len(sigmas))(*[float(s) for s in sigmas])
        shp = image.shape + (8,) if layout == INTERLEAVED else (8,) + image.shape
        out = np.empty((len(sigmas),) + shp, np.float32)
        self._chk(self._lib.ife_multi_emphysema_features(
            self._h, image.ctypes.data, _IMG_DT[image.dtype], mptr, mdt, C.byref(d), sig,
            len(sigmas), out.ctypes.data, layout))
        return out


    def emphysema_features_stream(self, image, mask, sigmas, spacing=(1.0, 1.0, 1.0), layout=INTERLEAVED):
        """Generator over the scales: one upload and prepass, every scale enqueued on every
        device, then one blocking fetch per scale (ife_multi_emphysema_features_begin / _fetch /
        _end) -- the scale loop of tools/ExtractFeatures.cxx:132-154 over several devices."""
        image = np.ascontiguousarray(image)
        if image.dtype not in _IMG_DT:
            image = image.astype(np.float32)
        mdt, mptr = U8, None
        if mask is not None:
            mask = np.ascontiguousarray(mask)
            mdt, mptr = _MSK_DT[mask.dtype], mask.ctypes.data
        d = _desc(image.shape, spacing)
        sig = (C.c_float * len(sigmas))(*[float(s) for s in sigmas])
        shp = image.shape + (8,) if layout == INTERLEAVED else (8,) + image.shape
        self._chk(self._lib.ife_multi_emphysema_features_begin(
            self._h, image.ctypes.data, _IMG_DT[image.dtype], mptr, mdt, C.byref(d), sig, len(sigmas), layout))
        try:
            for k in range(len(sigmas)):
                out = np.empty(shp, np.float32)
                self._chk(self._lib.ife_multi_emphysema_features_fetch(self._h, k, out.ctypes.data))
                yield out
        finally:
            self._chk(self._lib.ife_multi_emphysema_features_end(self._h))


class Samples:
    """``ife_samples``: one growing device-resident column per (scale, feature), the
    `samples` vector of tools/DetermineHistogramBinEdges_MultiScaleEigenvalueFeatures.cxx."""

    def __init__(self, ctx, n_columns):
        self._ctx, self._lib = ctx, ctx._lib
        self.n_columns = int(n_columns)
        h = C.c_void_p()
        ctx._chk(self._lib.ife_samples_create(ctx._h, self.n_columns, C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None) and getattr(self._ctx, "_h", None):
            self._lib.ife_samples_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def count(self, column=0):
        n = C.c_int64()
        self._ctx._chk(self._lib.ife_samples_count(self._h, int(column), C.byref(n)))
        return n.value

    def clear(self):
        self._ctx._chk(self._lib.ife_samples_clear(self._h))

    @staticmethod
    def _fg(foreground):
        fg = np.ascontiguousarray(foreground, np.uint32).ravel()
        return fg, (fg.ctypes.data if fg.size else None)

    def add_features(self, first_column, features, mask=None, foreground=(1,), indices=None,
                     layout=INTERLEAVED):
        """features: (nz, ny, nx, ncomp) interleaved or (ncomp, nz, ny, nx) planar, host."""
        f = np.ascontiguousarray(features, np.float32)
        ncomp = f.shape[-1] if layout == INTERLEAVED else f.shape[0]
        nvox = f.size // ncomp
        fg, fgp = self._fg(foreground)
        mptr, mdt, iptr, ni = None, U8, None, 0
        if indices is not None:
            indices = np.ascontiguousarray(indices, np.int64).ravel()
            iptr, ni = indices.ctypes.data, indices.size
        else:
            mask = np.ascontiguousarray(mask)
            mptr, mdt = mask.ctypes.data, _MSK_DT[mask.dtype]
        self._ctx._chk(self._lib.ife_samples_add_features(
            self._ctx._h, self._h, int(first_column), f.ctypes.data, layout, ncomp, mptr, mdt, nvox,
            fgp, fg.size, iptr, ni, MEM_HOST))

    def add_image(self, image, mask, sigmas, foreground=(1,), indices=None,
                  spacing=(1.0, 1.0, 1.0)):
        """One image of the tool's loop: features at every scale stay on the device."""
        image = np.ascontiguousarray(image)
        if image.dtype not in _IMG_DT:
            image = image.astype(np.float32)
        mask = np.ascontiguousarray(mask)
        d = _desc(image.shape, spacing)
        sig = (C.c_float * len(sigmas))(*[float(s) for s in sigmas])
        fg, fgp = self._fg(foreground)
        iptr, ni = None, 0
        if indices is not None:
            indices = np.ascontiguousarray(indices, np.int64).reshape(len(sigmas), -1)
            iptr, ni = indices.ctypes.data, indices.shape[1]
        self._ctx._chk(self._lib.ife_samples_add_image(
            self._ctx._h, self._h, image.ctypes.data, _IMG_DT[image.dtype], mask.ctypes.data,
            _MSK_DT[mask.dtype], C.byref(d), sig, len(sigmas), fgp, fg.size, iptr, ni, MEM_HOST))

    def add_image_device(self, image_ptr, image_dtype, mask_ptr, mask_dtype, shape_zyx, sigmas,
                         foreground=(1,), spacing=(1.0, 1.0, 1.0)):
        d = _desc(shape_zyx, spacing)
        sig = (C.c_float * len(sigmas))(*[float(s) for s in sigmas])
        fg, fgp = self._fg(foreground)
        self._ctx._chk(self._lib.ife_samples_add_image(
            self._ctx._h, self._h, C.c_void_p(image_ptr), image_dtype, C.c_void_p(mask_ptr),
            mask_dtype, C.byref(d), sig, len(sigmas), fgp, fg.size, None, 0, MEM_DEVICE))

    def sort(self):
        self._ctx._chk(self._lib.ife_samples_sort(self._ctx._h, self._h))

    def equalized_edges(self, nbins):
        """(n_columns, nbins-1) float32: one row per (scale, feature) as the tool writes them."""
        out = np.empty((self.n_columns, max(int(nbins) - 1, 0)), np.float32)
        self._ctx._chk(self._lib.ife_samples_equalized_edges(self._ctx._h, self._h, int(nbins),
                                                             out.ctypes.data))
        return out

    def column(self, column):
        out = np.empty(self.count(column), np.float32)
        self._ctx._chk(self._lib.ife_samples_read_column(self._ctx._h, self._h, int(column),
                                                         out.ctypes.data, out.size))
        return out
