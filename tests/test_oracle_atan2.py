"""Pin the oracle's fast_atan2f against the reference's own object code and golden vectors.

oracle/_ref/libref_fast_atan2f.so is /root/reference/multifm/fast_atan2f.c compiled unmodified
(oracle/Makefile).  tests/golden/atan2_ref.npz holds (y, x) -> bits it returned.
"""
import os

import numpy as np
import pytest


def _bits(a):
    return np.asarray(a, dtype=np.float32).view(np.uint32)


def test_oracle_matches_golden_vectors_from_reference(ora, golden_dir):
    g = np.load(os.path.join(golden_dir, "atan2_ref.npz"))
    yx, want = g["yx"], g["bits"]
    got = np.array([ora.lib().mfmo_fast_atan2f(float(y), float(x)) for y, x in yx], dtype=np.float32)
    assert np.array_equal(_bits(got), want), f"{int((_bits(got) != want).sum())} of {len(want)} differ"


def test_oracle_matches_reference_object_code_dense(ora):
    ref = ora.ref_atan2()
    if ref is None:
        pytest.skip("oracle/_ref not built (reference tree absent)")
    rng = np.random.RandomState(1234)
    n = 60000
    # the operands the discriminator produces: int32 values converted to float
    yx = np.concatenate([
        rng.randint(-(1 << 31), (1 << 31) - 1, size=(n // 3, 2)).astype(np.float32),
        rng.randint(-(1 << 16), 1 << 16, size=(n // 3, 2)).astype(np.float32),
        rng.randint(-300, 300, size=(n // 3, 2)).astype(np.float32),
    ])
    o = ora.lib()
    bad = 0
    for y, x in yx:
        a = np.float32(o.mfmo_fast_atan2f(float(y), float(x)))
        b = np.float32(ref.fast_atan2f(float(y), float(x)))
        bad += int(a.view(np.uint32) != b.view(np.uint32))
    assert bad == 0


def test_table_is_the_reference_table(ora):
    """T[i] read back through the reference function: for y = T-knot exact ratios the reference returns
    T[k] + (T[k+1]-T[k])*alpha with alpha in [0,1); at x=255,y=k it must land within the k-th segment."""
    ref = ora.ref_atan2()
    tbl = ora.atan_table()
    assert tbl.shape == (257,)
    assert tbl[0] == 0.0 and tbl[255] == tbl[256] == np.float32(7.853982e-01)
    assert np.all(np.diff(tbl[:256]) > 0)
    if ref is None:
        pytest.skip("oracle/_ref not built (reference tree absent)")
    # z == 1 hits index 255 with alpha 0: the reference returns pi/2 - T[255] for y >= x > 0
    v = np.float32(ref.fast_atan2f(7.0, 7.0))
    assert v == np.float32(np.float32(1.57079632679489661923) - tbl[255])
    # every knot: the returned angle lies inside [T[k-1], T[k+1]] (alpha rounding may move one segment)
    for k in range(1, 255):
        a = np.float32(ref.fast_atan2f(float(k), 255.0))
        assert tbl[k - 1] <= a <= tbl[k + 1]


def test_fused_variant_differs_by_at_most_one_pcm_lsb(ora):
    rng = np.random.RandomState(5)
    s = rng.randint(-(1 << 30), 1 << 30, size=(200000, 2)).astype(np.int32)
    sre, sim = np.ascontiguousarray(s[:, 0]), np.ascontiguousarray(s[:, 1])
    import ctypes as C
    a = np.zeros(len(s), np.int16)
    b = np.zeros(len(s), np.int16)
    p32 = C.POINTER(C.c_int32)
    ora.lib().mfmo_discriminate_batch(sre.ctypes.data_as(p32), sim.ctypes.data_as(p32), len(s), ora.p16(a), 0)
    ora.lib().mfmo_discriminate_batch(sre.ctypes.data_as(p32), sim.ctypes.data_as(p32), len(s), ora.p16(b), 1)
    d = np.abs(a.astype(np.int32) - b.astype(np.int32))
    assert d.max() <= 1
