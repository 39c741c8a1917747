"""The C++ multi-device engine (`ife_multi_*`, csrc/multi_capi.inc): the Z-slab decomposition
driven by one host thread with peer copies and HIP events.  The test box has one GPU, so the
device list names it several times -- every slab, chain item, stencil plane and event of the
W-device schedule runs, only the copies stay on one device.  Results must equal the
single-device path bit for bit (which is compared with the oracle elsewhere)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world,shape,dt,mdt,sigmas,spacing", [
    (1, (20, 24, 28), np.float32, np.uint8, [1.0, 2.0], (1, 1, 1)),
    (2, (33, 40, 36), np.float32, np.uint8, [1.0, 2.0, 4.0], (1, 1, 1)),
    (3, (29, 24, 70), np.int16, np.uint16, [1.5, 3.0], (0.7, 0.8, 1.25)),
    (4, (64, 130, 66), np.float32, np.uint8, [1.0, 2.0, 3.0, 4.0, 6.0], (1, 1, 1)),   # two scale groups
    (8, (37, 20, 24), np.float32, None, [1.0, 2.0, 4.0], (1, 1, 1)),                   # 5,5,5,5,5,4,4,4 planes; no mask
])
def test_multi_device_engine_equals_single_device(ife, synth, world, shape, dt, mdt, sigmas, spacing):
    img = synth.volume_i16(shape, 9) if dt == np.int16 else synth.volume_f32(shape, 9)
    mask = None
    if mdt is not None:
        mask = np.minimum(synth.mask_ellipsoids(shape), 1).astype(mdt)
        mask[:, :2, :] = 1
    with ife.Context(0) as c:
        c.set_option(ife.OPT_TRIG_MODE, 0)
        ref = c.emphysema_features(img, mask, sigmas, spacing)
        ref_planar = c.emphysema_features(img, mask, sigmas, spacing, layout=ife.PLANAR)
    with ife.Multi([0] * world) as m:
        m.set_option(ife.OPT_TRIG_MODE, 0)
        got = m.emphysema_features(img, mask, sigmas, spacing)
        np.testing.assert_array_equal(got, ref)
        again = m.emphysema_features(img, mask, sigmas, spacing)      # buffers and streams reused
        np.testing.assert_array_equal(again, ref)
        np.testing.assert_array_equal(m.emphysema_features(img, mask, sigmas, spacing, layout=ife.PLANAR),
                                      ref_planar)


def test_multi_device_engine_errors(ife, synth):
    with pytest.raises(ife.IfeError):
        ife.Multi([99])
    with ife.Multi([0, 0, 0]) as m:
        with pytest.raises(ife.IfeError) as e:
            m.emphysema_features(np.zeros((8, 8, 8), np.float32), None, [1.0])   # 8 planes over 3 devices
        assert e.value.code == ife.E_SIZE and "4 planes" in str(e.value)


@pytest.mark.gpu
def test_multi_device_streaming_equals_one_call(ife, synth):
    """ife_multi_emphysema_features_begin / _fetch / _end (one upload and prepass, every scale
    enqueued, one fetch per scale on a stream of its own) returns what the blocking call
    returns -- int16 input, uint16 labels, anisotropic spacing, five scales in two scale groups,
    uneven slabs -- and refuses calls out of sequence."""
    shape, spacing = (37, 40, 72), (0.7, 0.7, 1.0)
    img = synth.volume_i16(shape, 11)
    mask = np.minimum(synth.mask_ellipsoids(shape), 1).astype(np.uint16)
    sigmas = [1.0, 2.0, 3.0, 4.0, 6.0]
    with ife.Multi([0, 0, 0]) as m:
        m.set_option(ife.OPT_TRIG_MODE, 0)
        want = m.emphysema_features(img, mask, sigmas, spacing)
        got = list(m.emphysema_features_stream(img, mask, sigmas, spacing))
        assert len(got) == len(sigmas)
        for k, g in enumerate(got):
            np.testing.assert_array_equal(g.view(np.uint32), want[k].view(np.uint32))
        # a fetch without a begin, and a blocking call inside a streaming one
        out = np.empty(shape + (8,), np.float32)
        assert m._lib.ife_multi_emphysema_features_fetch(m._h, 0, out.ctypes.data) == ife.E_STATE
        gen = m.emphysema_features_stream(img, mask, sigmas[:2], spacing)
        next(gen)
        with pytest.raises(ife.IfeError):
            m.emphysema_features(img, mask, sigmas[:1], spacing)
        gen.close()   # runs _end
        np.testing.assert_array_equal(m.emphysema_features(img, mask, sigmas[:1], spacing)[0], want[0])
    with ife.Context(0) as c:
        c.set_option(ife.OPT_TRIG_MODE, 0)
        ref = c.emphysema_features(img, mask, sigmas, spacing)
    np.testing.assert_array_equal(want.view(np.uint32), ref.view(np.uint32))
