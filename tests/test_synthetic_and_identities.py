"""CPU tests: the synthetic input generator is deterministic, and the division identity
the HIP solver relies on holds for every float."""
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def test_synthetic_volume_is_pinned(synth):
    v = synth.volume_f32((3, 4, 5), synth.SEED_CONFIG[3])
    assert v.dtype == np.float32 and v.shape == (3, 4, 5)
    # integer hash + integer structure term: identical on every host
    assert [float.hex(float(x)) for x in v[0, 0, :3]] == [
        float.hex(float(x)) for x in synth.volume_f32((1, 1, 3), synth.SEED_CONFIG[3])[0, 0]]
    a = synth.volume_f32((8, 6, 7), 42)
    b = synth.volume_f32((3, 6, 7), 42, z0=5)
    np.testing.assert_array_equal(a[5:], b)           # slabs of one volume agree
    assert -3100 < a.min() < a.max() < 3100
    i16 = synth.volume_i16((4, 4, 4), 7)
    assert i16.dtype == np.int16 and i16.min() >= -1024 and i16.max() <= 3071


def test_mask_has_two_labels_and_reasonable_foreground(synth):
    m = synth.mask_ellipsoids((40, 40, 40))
    assert set(np.unique(m)) == {0, 1, 2}
    assert 0.1 < (m > 0).mean() < 0.4
    np.testing.assert_array_equal(synth.mask_ellipsoids((10, 40, 40), z0=20, nz_total=40), m[20:30])


def test_division_by_3_and_6_identity_is_exhaustive():
    """eigen_device.hpp div_by_const == IEEE x/3, x/6 for all 2^32 floats."""
    flags = open("/proc/cpuinfo").read()
    if " fma" not in flags:
        pytest.skip("host CPU has no FMA instruction")
    exe = "/tmp/ife_div_const_exhaustive"
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fopenmp", "-mfma", "-o", exe,
                           os.path.join(HERE, "csrc", "div_const_exhaustive.c"), "-lm"])
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout
    assert "d=3 mismatches=0" in r.stdout and "d=6 mismatches=0" in r.stdout
