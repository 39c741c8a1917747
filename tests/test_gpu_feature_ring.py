"""The two forms of the feature kernel (feature_kernels.hpp): planes staged through registers
(IFE_OPT_FEAT_RING=0) and planes by LDS-DMA into a ring with counted waits (=1, default where
the source is one float field and the mask is 1 or 2 bytes wide).  Both call the same
arithmetic function, so every output must have the same bits; the oracle comparisons of the
other test files run on the default form.  Shapes are chosen around what the ring form
counts on: rows that end inside a 64-voxel tile, tile rows below the volume, z-chunks shorter
than the prefetch distance, masks whose rows are not dword multiples (the ring form must
decline those), every output mode and layout."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def both(ctx, ife, fn):
    ctx.set_option(ife.OPT_FEAT_RING, 1)
    try:
        a = fn()
        ctx.set_option(ife.OPT_FEAT_RING, 0)
        b = fn()
    finally:
        ctx.set_option(ife.OPT_FEAT_RING, 1)
    return a, b


def same_bits(a, b):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    assert a.shape == b.shape and a.dtype == b.dtype
    np.testing.assert_array_equal(a.view(np.uint32), b.view(np.uint32))


SHAPES = [(9, 10, 70), (33, 36, 40), (5, 4, 132), (70, 9, 4), (21, 19, 68), (12, 64, 64),
          (7, 8, 200), (6, 17, 63)]


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("zchunk", [64, 3, 1])
@pytest.mark.parametrize("trig", [0, 2])
def test_emphysema_features_ring_equals_staged(ctx, ife, synth, shape, zchunk, trig):
    img = synth.volume_f32(shape, 77)
    labels = synth.mask_ellipsoids(shape)
    ctx.set_option(ife.OPT_ZCHUNK, zchunk)
    ctx.set_option(ife.OPT_TRIG_MODE, trig)
    try:
        for mask in (np.minimum(labels, 1).astype(np.uint8), np.minimum(labels, 1).astype(np.uint16), None):
            for layout in (ife.INTERLEAVED, ife.PLANAR):
                a, b = both(ctx, ife, lambda: ctx.emphysema_features(img, mask, [1.0, 2.0], layout=layout))
                same_bits(a, b)
        a, b = both(ctx, ife, lambda: ctx.emphysema_features(
            img, np.minimum(labels, 1).astype(np.uint8), [1.5], spacing=(0.7, 0.8, 1.25)))
        same_bits(a, b)
    finally:
        ctx.set_option(ife.OPT_ZCHUNK, 64)
        ctx.set_option(ife.OPT_TRIG_MODE, 0)


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("zchunk", [64, 2])
def test_tool_bodies_ring_equals_staged(ctx, ife, synth, shape, zchunk):
    img = synth.volume_f32(shape, 5)
    mask = np.minimum(synth.mask_ellipsoids(shape), 1).astype(np.uint8)
    ctx.set_option(ife.OPT_ZCHUNK, zchunk)
    try:
        for spacing in ((1, 1, 1), (0.7, 0.8, 1.25)):
            for layout in (ife.INTERLEAVED, ife.PLANAR):
                same_bits(*both(ctx, ife, lambda: ctx.hessian3d(img, spacing, layout=layout)))
                same_bits(*both(ctx, ife, lambda: ctx.fd_hessian_features(img, mask, spacing, layout=layout)))
                same_bits(*both(ctx, ife, lambda: ctx.fd_hessian_features(img, None, spacing, layout=layout)))
            same_bits(*both(ctx, ife, lambda: ctx.gradient_magnitude(img, spacing)))
    finally:
        ctx.set_option(ife.OPT_ZCHUNK, 64)


@pytest.mark.parametrize("shape", [(33, 36, 40), (9, 10, 132), (21, 19, 68), (6, 17, 63)])
@pytest.mark.parametrize("zchunk", [64, 3])
def test_sampled_columns_ring_equals_staged(ctx, ife, synth, shape, zchunk):
    """Row f1's fused sampling (FEAT_SAMPLES8): the eight features of the sampled voxels go
    straight into sample columns; unsampled voxels are dropped by the store's descriptor in
    the ring form, by EXEC in the staged one."""
    img = synth.volume_f32(shape, 9)
    labels = synth.mask_ellipsoids(shape)
    ctx.set_option(ife.OPT_ZCHUNK, zchunk)

    def run():
        s = ctx.samples(16)
        try:
            s.add_image(img, labels, [1.0, 2.0], foreground=(1, 2))
            n = s.count(0)
            return np.stack([s.column(c) for c in range(16)]), n
        finally:
            s.close()
    try:
        (a, na), (b, nb) = both(ctx, ife, run)
    finally:
        ctx.set_option(ife.OPT_ZCHUNK, 64)
    assert na == nb == int((labels > 0).sum())
    same_bits(a, b)
