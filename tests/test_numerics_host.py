"""The kernel's scalar numerics (tsl-sdr_amd/csrc/mfm_numerics.h), compiled for the host inside
libmultifm_hip.so as mfm_hosttwin_*, against the oracle.  No GPU needed: these are the same source
lines the device code is built from."""
import ctypes as C
from concurrent.futures import ThreadPoolExecutor

import numpy as np


def test_atan_table_matches_oracle_and_pinned_hash(pkg, ora):
    lib = pkg.load_library()
    t = (C.c_float * 257)()
    lib.mfm_hosttwin_atan_table(t)
    assert np.array_equal(np.frombuffer(t, dtype=np.uint32), ora.atan_table().view(np.uint32))
    assert lib.mfm_hosttwin_atan_table_ok() == 1


def test_r14_matches_oracle(pkg, ora):
    lib = pkg.load_library()
    rng = np.random.RandomState(3)
    vals = list(rng.randint(-(1 << 31), (1 << 31) - 1, size=20000)) + [0, 1, -1, 8191, 8192, 8193, -8192, -8193,
                                                                        (1 << 31) - 1, -(1 << 31), 1 << 29]
    for v in vals:
        assert lib.mfm_hosttwin_r14(int(v)) == ora.lib().mfmo_r14(int(v))


def test_angle_to_pcm_is_exact_for_every_float_in_0_pi(pkg, ora):
    """(int)fmaf(m, hi, m*lo) == (int16)(float)(((double)m / M_PI) * 16384.0) (fm_demod.c:71-72) for all
    1 078 530 012 floats m in [0, (float)pi] - the only values |fast_atan2f| can take."""
    lib, o = pkg.load_library(), ora.lib()
    top = int(np.float32(3.14159265358979323846).view(np.uint32))
    step = 1 << 24
    starts = list(range(0, top + 1, step))

    def work(s):
        n = min(step, top + 1 - s)
        a = np.empty(n, np.int16)
        b = np.empty(n, np.int16)
        lib.mfm_hosttwin_pcm_range(s, n, a.ctypes.data_as(C.POINTER(C.c_int16)))
        o.mfmo_phi_to_pcm_range(s, n, b.ctypes.data_as(C.POINTER(C.c_int16)))
        return int((a != b).sum())

    with ThreadPoolExecutor(max_workers=8) as ex:
        bad = sum(ex.map(work, starts))
    assert bad == 0


def _disc_both(pkg, ora, sre, sim):
    lib, o = pkg.load_library(), ora.lib()
    sre = np.ascontiguousarray(sre, dtype=np.int32)
    sim = np.ascontiguousarray(sim, dtype=np.int32)
    a = np.empty(len(sre), np.int16)
    b = np.empty(len(sre), np.int16)
    p32 = C.POINTER(C.c_int32)
    lib.mfm_hosttwin_discriminate_batch(sre.ctypes.data_as(p32), sim.ctypes.data_as(p32), len(sre),
                                        a.ctypes.data_as(C.POINTER(C.c_int16)))
    o.mfmo_discriminate_batch(sre.ctypes.data_as(p32), sim.ctypes.data_as(p32), len(sre), ora.p16(b), 0)
    return a, b


def test_discriminator_matches_oracle_random(pkg, ora):
    rng = np.random.RandomState(17)
    parts = []
    for bits in (3, 8, 14, 20, 26, 31):
        lim = 1 << bits
        parts.append(rng.randint(-lim, lim - 1, size=(400000, 2)))
    s = np.concatenate(parts).astype(np.int32)
    a, b = _disc_both(pkg, ora, s[:, 0], s[:, 1])
    assert np.array_equal(a, b), f"{int((a != b).sum())} of {len(a)} differ"


def test_discriminator_matches_oracle_edges(pkg, ora):
    e = [0, 1, -1, 2, -2, 255, -255, 256, 32767, -32768, 65535, (1 << 24) - 1, 1 << 24, (1 << 24) + 1,
         (1 << 30), -(1 << 30), (1 << 31) - 1, -(1 << 31), -(1 << 31) + 1, 1073741823, 16777217, 33554433]
    sre, sim = np.meshgrid(np.array(e, np.int64), np.array(e, np.int64))
    a, b = _disc_both(pkg, ora, sre.reshape(-1), sim.reshape(-1))
    assert np.array_equal(a, b)
    # exact table knots and the small-angle threshold: y/x = k/255
    k = np.arange(-255, 256)
    for scale in (1, 3, 1000, 1 << 16):
        a, b = _disc_both(pkg, ora, np.full(k.shape, 255 * scale), k * scale)
        assert np.array_equal(a, b)
        a, b = _disc_both(pkg, ora, k * scale, np.full(k.shape, -255 * scale))
        assert np.array_equal(a, b)
    # dense small grid: every (s_re, s_im) in [-300, 300]^2
    g = np.arange(-300, 301)
    sre, sim = np.meshgrid(g, g)
    a, b = _disc_both(pkg, ora, sre.reshape(-1), sim.reshape(-1))
    assert np.array_equal(a, b)


def test_byte_plane_sums_of_8bit_input_equal_the_widened_sums(ora):
    """The arithmetic the IN8 kernel forms rest on (mfm_kernel_v3.hip), in numpy: with the taps split
    as the engine splits them (W = 256 Wh + Wl, both int8) and the byte taken as int8 s (after ^ 0x80 for the RTL-SDR),
    bits [sh + 15 : sh] of (sum Wh s << 8) + sum Wl s + K are the first rounding of the reference's wrapping int32 sum over
    the WIDENED samples (filter/complex.h:30-34 on rtl_sdr_if.c:146-148 / file_if.c:66-157), for K = (beta sum W + 8192) /
    alpha and sh = 14 - log2 alpha.  Full-range taps and bytes, sums that wrap."""
    rng = np.random.RandomState(12)
    for fmt, xor, alpha, beta, sh in ((1, 0, 1, 0, 14), (2, 0, 1, -127, 14), (3, 0x80, 128, 128, 7)):
        for trial in range(200):
            k = int(rng.choice([4, 64, 256, 1024]))  # an even number of IQ pairs: no cu8 odd-tail quirk (file_if.c:146-150)
            lim = int(rng.choice([120, 3000, 32639]))
            w = rng.randint(-lim, lim + 1, size=k).astype(np.int64)
            raw = rng.randint(0, 256, size=k).astype(np.uint8)
            if trial == 0:
                w[:] = 32639
                raw[:] = 255
            if trial == 1:
                w[:] = -32639
                raw[:] = 128 if fmt != 3 else 0
            x = ora.unpack_bytes(raw, fmt).astype(np.int64)          # the reference's widening, sample by sample
            assert x.size == k
            ref = int((w * x).sum()) & 0xFFFFFFFF                    # wrapping int32 sum (filter/direct_fir.c)
            want = ((ref + 8192) & 0xFFFFFFFF) >> 14 & 0xFFFF       # round_q30_q15 + int16 truncation
            wl = ((w & 0xFF) ^ 0x80) - 0x80                          # (int8)(w & 0xff)
            wh = (w - wl) >> 8
            assert np.all(np.abs(wh) <= 128) and np.all(w == 256 * wh + wl)
            s = ((raw.astype(np.int64) ^ xor) ^ 0x80) - 0x80         # the byte as int8
            assert np.array_equal(alpha * s + beta, x)
            sw = int(w.sum())
            kk = {1: 8192, 2: 8192 - 127 * sw, 3: sw + 64}[fmt]
            assert kk * alpha == beta * sw + 8192
            t = ((int((wh * s).sum()) << 8) + int((wl * s).sum()) + kk) & 0xFFFFFFFF
            assert (t >> sh) & 0xFFFF == want, (fmt, trial)


def test_division_with_one_residual_step_enumerated(tmp_path):
    """mfm_div_unit (tsl-sdr_amd/csrc/mfm_numerics.h): reciprocal, one Newton step, quotient estimate, ONE residual step.
    tools/div_proof.c enumerates every pair of significands whose quotient lies close enough to a midpoint of two floats for
    the final rounding to go wrong (46.5 M pairs over all 2^23 divisors) and runs the float sequence on each: with the
    correctly rounded reciprocal none goes wrong (Markstein's theorem, by enumeration); with reciprocals up to two ulp off
    exactly three pairs do - the ones the engine's division self-test runs on the device at commit."""
    import os
    import subprocess
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "div_proof"
    r = subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-o", str(exe), os.path.join(ROOT, "tools", "div_proof.c"), "-lm",
                        "-lpthread"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe), "1", "8", "-", "rn"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and " 0 wrong; 5e7 random pairs: 0 wrong" in r.stdout, r.stdout[-500:]
    r = subprocess.run([str(exe), "1", "8"], capture_output=True, text=True, timeout=600)
    bad = sorted(set(tuple(int(t.split("=")[1]) for t in ln.split()[1:3]) for ln in r.stdout.splitlines() if ln.startswith("FAIL")))
    assert bad == [(8388608, 16777215), (13981011, 16777213), (15099490, 16777211)], r.stdout[-800:]
    src = open(os.path.join(ROOT, "tsl-sdr_amd", "csrc", "mfm_engine.hip")).read()
    for a, b in bad:
        assert "{ %d, %d }" % (a, b) in src, "the commit-time division self-test must run this quotient"
