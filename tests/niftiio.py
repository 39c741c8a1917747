"""Minimal NIfTI-1 (.nii / .nii.gz) reader/writer for the tool tests (numpy + gzip only)."""
import gzip
import struct

import numpy as np

_CODES = {np.dtype(np.uint8): 2, np.dtype(np.int16): 4, np.dtype(np.uint16): 512,
          np.dtype(np.float32): 16, np.dtype(np.float64): 64}
_DTYPES = {v: k for k, v in _CODES.items()}


GEOM_FIELDS = (("qform_code", "<h", 252), ("sform_code", "<h", 254), ("quatern", "<3f", 256),
               ("qoffset", "<3f", 268), ("srow_x", "<4f", 280), ("srow_y", "<4f", 296),
               ("srow_z", "<4f", 312), ("qfac", "<f", 76))


def read_geometry(path):
    """The world-geometry fields of the header, as raw tuples."""
    raw = gzip.open(path, "rb").read(352) if path.endswith(".gz") else open(path, "rb").read(352)
    return {k: struct.unpack_from(f, raw, o) for k, f, o in GEOM_FIELDS}


def write(path, vol_zyx, spacing_xyz=(1.0, 1.0, 1.0), geometry=None, patch=None):
    """geometry: {field: tuple} of GEOM_FIELDS to set; patch: [(fmt, offset, values...)] raw
    header edits (malformed-header tests)."""
    vol = np.ascontiguousarray(vol_zyx)
    nz, ny, nx = vol.shape
    h = bytearray(348)
    struct.pack_into("<i", h, 0, 348)
    struct.pack_into("<8h", h, 40, 3, nx, ny, nz, 1, 1, 1, 1)
    struct.pack_into("<hh", h, 70, _CODES[vol.dtype], vol.dtype.itemsize * 8)
    struct.pack_into("<8f", h, 76, 1.0, *[float(s) for s in spacing_xyz], 0, 0, 0, 0)
    struct.pack_into("<f", h, 108, 352.0)
    struct.pack_into("<f", h, 112, 1.0)
    h[344:348] = b"n+1\0"
    for k, f, o in GEOM_FIELDS:
        if geometry and k in geometry:
            struct.pack_into(f, h, o, *geometry[k])
    for f, o, *vals in (patch or []):
        struct.pack_into(f, h, o, *vals)
    data = bytes(h) + b"\0\0\0\0" + vol.tobytes()
    if path.endswith(".gz"):
        with gzip.open(path, "wb", compresslevel=1) as f:
            f.write(data)
    else:
        with open(path, "wb") as f:
            f.write(data)


def read(path):
    raw = gzip.open(path, "rb").read() if path.endswith(".gz") else open(path, "rb").read()
    assert struct.unpack_from("<i", raw, 0)[0] == 348 and raw[344:347] == b"n+1"
    dim = struct.unpack_from("<8h", raw, 40)
    code = struct.unpack_from("<h", raw, 70)[0]
    pix = struct.unpack_from("<8f", raw, 76)
    off = int(struct.unpack_from("<f", raw, 108)[0])
    nx, ny, nz = dim[1], dim[2], dim[3]
    dt = _DTYPES[code]
    vol = np.frombuffer(raw, dt, nx * ny * nz, off).reshape(nz, ny, nx)
    return vol, (pix[1], pix[2], pix[3])
