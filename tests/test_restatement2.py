"""The oracle (oracle/mfm_oracle.c) against a second, independently written restatement of SURVEY.md Appendix A
(tests/restatement2.py: numpy / Python integers, from the reference's sources) and against the configuration-only known
answers the survey's probe of the compiled reference recorded (tests/golden/survey_probe.json).  Neither is a compiled
reference - parity stays "unpinned" - but a misreading of direct_fir.c / fm_demod.c would now have to be made twice, by two
different constructions, to pass."""
import json
import os

import numpy as np
import pytest

import restatement2 as r2

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _probe():
    return json.load(open(os.path.join(GOLDEN, "survey_probe.json")))


def test_both_restatements_reproduce_the_survey_probe_values(ora):
    p = _probe()
    for oc in p["output_counts"]:
        want = oc["outputs"]
        assert ora.expected_outputs(oc["samples"], oc["taps"], oc["decimation"]) == want
        assert (oc["samples"] - oc["taps"]) // oc["decimation"] + 1 == want
    for rc in p["rotator"]:
        fs, D = rc["sample_rate_hz"], rc["decimation"]
        inc2 = r2.rot_increment(rc["offset_hz"], fs, D)
        inc1 = tuple(int(v) for v in ora.rot_incr(rc["offset_hz"], fs, D))
        assert inc1 == inc2
        if "increment" in rc:
            assert list(inc2) == rc["increment"]
        rr, ri = 16384, 0
        for _ in range(rc["steps"]):
            nr = int(r2.r14(r2._wrap32(rr * inc2[0] - ri * inc2[1])))
            ni = int(r2.r14(r2._wrap32(rr * inc2[1] + ri * inc2[0])))
            rr, ri = nr, ni
        assert int(round((rr * rr + ri * ri) ** 0.5)) == rc["magnitude_after"]
        ch = ora.Channel(np.ones(128, np.int16), np.zeros(128, np.int16), D, inc1)
        ch.skip_outputs(rc["steps"])
        assert tuple(int(v) for v in ch.rot()) == (rr, ri)
        ch.close()
    for off in p["exact_rotator_offsets_hz"]["offsets"]:
        inc = r2.rot_increment(off, p["exact_rotator_offsets_hz"]["sample_rate_hz"], p["exact_rotator_offsets_hz"]["decimation"])
        assert sorted(abs(v) for v in inc) == [0, 16384]


def test_tap_rotation_and_table_agree(pkg, ora):
    assert np.array_equal(r2.atan_table().view(np.uint32), ora.atan_table().view(np.uint32))
    for name in ("multifm_1ch", "cfg2_64ch", "pocsag_rtlsdr", "cfg5_airspy"):
        fs, decim, taps, offs, gains = pkg.synth.plan(name, nr_channels=None if name != "cfg5_airspy" else 6)
        for o, g in list(zip(offs, gains))[:6]:
            a = r2.taps(taps, int(o), fs, float(g))
            b = ora.make_taps(taps, int(o), fs, float(g))
            assert np.array_equal(a[0], b[0].astype(np.int64)) and np.array_equal(a[1], b[1].astype(np.int64)), (name, o)
            assert r2.rot_increment(int(o), fs, decim) == tuple(int(v) for v in ora.rot_incr(int(o), fs, decim))


def test_golden_vector_through_the_second_restatement(golden_dir):
    g = np.load(os.path.join(golden_dir, "path_oracle.npz"))
    decim = int(g["decim"])
    for c in range(len(g["offsets"])):
        pcm, q = r2.channel(g["iq"], g["cre"][c], g["cim"][c], decim, g["incr"][c])
        assert np.array_equal(pcm, g["pcm"][c]), c
        assert np.array_equal(q, g["filt_iq"][c].reshape(-1, 2)), c


@pytest.mark.parametrize("case", ["full_scale_wraps", "decaying_rotator", "exact_rotators", "zero_increment"])
def test_streams_agree_between_the_two_restatements(pkg, ora, case):
    rng = np.random.RandomState(11)
    fs, decim = 2400000, 96
    lpf = pkg.synth.design_lpf(128, 12500.0, fs)
    if case == "full_scale_wraps":
        # taps near full scale and full-scale input: the int32 accumulator wraps many times per output
        n = decim * 1500 + 128
        iq = rng.randint(-32768, 32768, size=(n, 2)).astype(np.int16)
        chans = [(rng.randint(-32639, 32640, 128), rng.randint(-32639, 32640, 128), (11585, -11585)),
                 (np.full(128, 32639), np.full(128, -32639), (16100, 3000))]
    elif case == "decaying_rotator":
        n = decim * 60000 + 128   # 101 kHz: the rotator's magnitude falls from 16384 to a few thousand on the way
        iq = pkg.synth.synth_iq(n, fs, [101000, 777], seed=3, noise=3000)
        chans = [(*r2.taps(lpf, o, fs, 1.0), r2.rot_increment(o, fs, decim)) for o in (101000, 777)]
    elif case == "exact_rotators":
        n = decim * 3000 + 500
        iq = rng.randint(-32768, 32768, size=(n, 2)).astype(np.int16)   # -32768 reaches the sign changes
        chans = [(*r2.taps(lpf, o, fs, 2.0), r2.rot_increment(o, fs, decim)) for o in (25000, 12500, 6250, -6250)]
    else:
        n = decim * 2000 + 128
        iq = pkg.synth.synth_iq(n, fs, [0], seed=4)
        chans = [(*r2.taps(lpf, 0, fs, 1.0), (0, 0))]
    for cre, cim, incr in chans:
        pcm2, q2 = r2.channel(iq, cre, cim, decim, incr)
        ch = ora.Channel(np.asarray(cre, np.int16), np.asarray(cim, np.int16), decim, incr)
        pcm1, q1 = ch.feed(iq)
        ch.close()
        assert pcm1.shape == pcm2.shape and np.array_equal(pcm1, pcm2), (case, incr)
        assert np.array_equal(q1.reshape(-1, 2), q2), (case, incr)
    if case == "decaying_rotator":
        mag = np.hypot(*[float(v) for v in (q2[-1])])
        assert pcm2[0] == 0   # fm_demod.c: the previous sample starts at zero


def test_discriminator_agrees_on_products(ora):
    import ctypes as C
    rng = np.random.RandomState(2)
    s_re = np.concatenate([rng.randint(-2**31, 2**31, size=200000, dtype=np.int64), [0, 0, 1, -1, 5, 2**31 - 1, -2**31],
                           rng.randint(-300, 300, size=50000)]).astype(np.int32)
    s_im = np.concatenate([rng.randint(-2**31, 2**31, size=200000, dtype=np.int64), [0, 7, 0, -1, 5, -2**31, 2**31 - 1],
                           rng.randint(-300, 300, size=50000)]).astype(np.int32)
    want = np.zeros(s_re.size, np.int16)
    i32p, i16p = C.POINTER(C.c_int32), C.POINTER(C.c_int16)
    ora.lib().mfmo_discriminate_batch(s_re.ctypes.data_as(i32p), s_im.ctypes.data_as(i32p), s_re.size, want.ctypes.data_as(i16p), 0)
    phi = r2.fast_atan2f(s_im.astype(np.float32), s_re.astype(np.float32))
    got = np.trunc(((phi.astype(np.float64) / np.pi) * 16384.0).astype(np.float32)).astype(np.int64).astype(np.int16)
    assert np.array_equal(got, want)
