"""Deterministic synthetic volumes: integer hash only, no libm,
so that every host produces bit-identical inputs.

  u(i)   = splitmix64(seed + i)
  fp32   = ((u >> 40) & 0xFFFFFF) * 2^-24 * 2000 - 1000      uniform noise in [-1000, 1000)
         + ((3x^2 + 5y^2 + 7z^2 + xy + 2yz) mod 4096) - 2048   integer structure term
  int16  = clamp(round-toward-zero of the same expression, -1024, 3071)
  mask   = union of two axis-aligned ellipsoids with labels 1 and 2 (about a quarter of
           the voxels), which callers clamp to {0, 1} as ExtractFeatures.cxx:99-104 does
"""
import numpy as np

SEED_CONFIG = {1: 0x1FE00001, 2: 0x1FE00002, 3: 0x1FE00003, 4: 0x1FE00003, 5: 0x1FE00005}


def splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
    z = x
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def _structure(nz, ny, nx, z0=0):
    z = (np.arange(nz, dtype=np.int64) + z0)[:, None, None]
    y = np.arange(ny, dtype=np.int64)[None, :, None]
    x = np.arange(nx, dtype=np.int64)[None, None, :]
    return ((3 * x * x + 5 * y * y + 7 * z * z + x * y + 2 * y * z) % 4096) - 2048


def volume_f32(shape_zyx, seed, z0=0, nz_total=None):
    """Planes [z0, z0+nz) of the synthetic float32 volume of full depth nz_total."""
    nz, ny, nx = shape_zyx
    out = np.empty((nz, ny, nx), np.float32)
    with np.errstate(over="ignore"):
        for k in range(nz):  # plane by plane keeps the temporaries small
            i = (np.arange(ny * nx, dtype=np.uint64) + np.uint64((z0 + k) * ny * nx)
                 + np.uint64(seed))
            u = splitmix64(i)
            noise = ((u >> np.uint64(40)) & np.uint64(0xFFFFFF)).astype(np.float32)
            noise = noise * np.float32(2.0 ** -24) * np.float32(2000.0) - np.float32(1000.0)
            out[k] = noise.reshape(ny, nx) + _structure(1, ny, nx, z0 + k)[0].astype(np.float32)
    return out


def volume_i16(shape_zyx, seed, z0=0):
    v = volume_f32(shape_zyx, seed, z0)
    return np.clip(np.trunc(v), -1024, 3071).astype(np.int16)


def mask_ellipsoids(shape_zyx, z0=0, nz_total=None):
    """Labels 0/1/2: two ellipsoids covering about a quarter of the volume."""
    nz, ny, nx = shape_zyx
    nzt = nz_total or nz
    z = (np.arange(nz, dtype=np.int64) + z0)[:, None, None]
    y = np.arange(ny, dtype=np.int64)[None, :, None]
    x = np.arange(nx, dtype=np.int64)[None, None, :]

    def inside(cx, cy, cz, rx, ry, rz):
        # integer inequality: (x-cx)^2 ry^2 rz^2 + ... <= rx^2 ry^2 rz^2
        return ((x - cx) ** 2 * (ry * rz) ** 2 + (y - cy) ** 2 * (rx * rz) ** 2
                + (z - cz) ** 2 * (rx * ry) ** 2) <= (rx * ry * rz) ** 2

    m = np.zeros((nz, ny, nx), np.uint8)
    r1 = (max(nx * 22 // 100, 1), max(ny * 34 // 100, 1), max(nzt * 40 // 100, 1))
    r2 = (max(nx * 20 // 100, 1), max(ny * 30 // 100, 1), max(nzt * 36 // 100, 1))
    m[inside(nx * 28 // 100, ny // 2, nzt // 2, *r1)] = 1
    m[inside(nx * 72 // 100, ny // 2, nzt // 2, *r2)] = 2
    return m
