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
NUM_FEATURES = 8
FEATURE_NAMES = ("GaussianBlur", "GradientMagnitude", "Eigenvalue1", "Eigenvalue2",
                 "Eigenvalue3", "LaplacianOfGaussian", "GaussianCurvature", "FrobeniusNorm")

# every symbol include/ife_hip.h declares
EXPORTS = (
    "ife_abi_version", "ife_ctx_create", "ife_ctx_destroy", "ife_last_error",
    "ife_ctx_set_stream", "ife_ctx_set_option", "ife_ctx_reserve", "ife_ctx_synchronize",
    "ife_eigenvalues", "ife_eigenvalue_features", "ife_hessian3d", "ife_gradient_magnitude",
    "ife_normalized_gaussian_convolution", "ife_emphysema_features",
    "ife_fd_hessian_features", "ife_fd_gradient_features", "ife_mask_image_f64",
    "ife_get_kernel_times", "ife_reset_kernel_times",
    "ife_stage_prepare", "ife_stage_recursive_gaussian", "ife_stage_recursive_gaussian_batch",
    "ife_stage_features",
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
    lib.ife_emphysema_features.argtypes = [vp, vp, i32, vp, i32, vd, C.POINTER(C.c_float), i32,
                                           f32p, i32, i32]
    lib.ife_fd_hessian_features.argtypes = [vp, vp, i32, vp, i32, vd, f32p, i32, i32]
    lib.ife_fd_gradient_features.argtypes = [vp, f32p, f32p, vd, f32p, i32]
    lib.ife_mask_image_f64.argtypes = [vp, vp, vp, C.c_double, i64, vp, i32]
    lib.ife_stage_prepare.argtypes = [vp, vp, i32, vp, i32, vd, i32, vp, vp]
    lib.ife_stage_recursive_gaussian.argtypes = [vp, vp, vp, vd, i32, C.c_double]
    lib.ife_stage_recursive_gaussian_batch.argtypes = [vp, i32, C.POINTER(vp), C.POINTER(vp), vd,
                                                       i32, C.POINTER(C.c_double), i32]
    lib.ife_stage_features.argtypes = [vp, vp, vp, vp, i32, vd, i32, i32, vp, i32]
    lib.ife_get_kernel_times.argtypes = [vp, C.POINTER(KernelTime), i32]
    lib.ife_reset_kernel_times.argtypes = [vp]
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
