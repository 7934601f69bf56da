"""Oracle self-consistency and the known answers the survey's probe of the compiled reference recorded
(SURVEY.md facts 3, 4, 6 and section 8): output counts, chunking independence, rotator decay."""
import os

import numpy as np
import pytest


def _plan(pkg, ora, offs, fs=2400000, decim=96, ntaps=128, gains=None):
    taps = pkg.synth.design_lpf(ntaps, 12500.0, fs)
    gains = gains or [1.0] * len(offs)
    cre = np.stack([ora.make_taps(taps, o, fs, g)[0] for o, g in zip(offs, gains)])
    cim = np.stack([ora.make_taps(taps, o, fs, g)[1] for o, g in zip(offs, gains)])
    incr = np.stack([ora.rot_incr(o, fs, decim) for o in offs])
    return cre, cim, incr


def test_output_counts_match_reference_probe(ora):
    # N_out(2^20) = 10 922 for T=128, D=96 and 2 621 for T=512, D=400
    assert ora.expected_outputs(1 << 20, 128, 96) == 10922
    assert ora.expected_outputs(1 << 20, 512, 400) == 2621
    assert ora.expected_outputs(127, 128, 96) == 0
    assert ora.expected_outputs(128, 128, 96) == 1


def test_rotator_increment_and_decay_match_reference_probe(ora):
    # offset 3125 Hz, D=96, fs=2.4 MHz -> incr (11585,-11585); |rot| 16384 -> 16382 after 4k
    incr = ora.rot_incr(3125, 2400000, 96)
    assert tuple(int(v) for v in incr) == (11585, -11585)
    r = np.array([16384, 0], np.int16)
    for _ in range(4096):
        ora.lib().mfmo_rot_step(ora.p16(r[0:1]), ora.p16(r[1:2]), int(incr[0]), int(incr[1]))
    assert int(round(np.hypot(float(r[0]), float(r[1])))) == 16382
    # offset 101 kHz: 16384 -> 4574 after 128 k outputs
    incr = ora.rot_incr(101000, 2400000, 96)
    r = np.array([16384, 0], np.int16)
    for _ in range(128 * 1024):
        ora.lib().mfmo_rot_step(ora.p16(r[0:1]), ora.p16(r[1:2]), int(incr[0]), int(incr[1]))
    assert int(round(np.hypot(float(r[0]), float(r[1])))) == 4574
    # exact quarter-turn increments stay exact
    for off in (37500, -1181250, 25000, 6250):
        i = ora.rot_incr(off, 2400000, 96)
        assert sorted(abs(int(v)) for v in i) == [0, 16384]


def test_first_pcm_is_zero_and_stream_is_chunking_independent(pkg, ora):
    offs = [101000, -433219]
    cre, cim, incr = _plan(pkg, ora, offs)
    n = 96 * 2000 + 128 + 17
    iq = pkg.synth.synth_iq(n, 2400000, offs, seed=2)
    whole, whole_q = ora.run_channels(iq, cre, cim, incr, 96, want_iq=True)
    assert whole.shape[1] == ora.expected_outputs(n, 128, 96)
    assert np.all(whole[:, 0] == 0)  # fm_demod.c: previous sample starts at 0 -> atan2(0,0) = 0
    rng = np.random.RandomState(0)
    for sizes in ([4096], [1000], [131072], [1, 127, 128, 129, 4000], list(rng.randint(1, 5000, size=64))):
        for c in range(len(offs)):
            ch = ora.Channel(cre[c], cim[c], 96, incr[c])
            outs, qs, pos, k = [], [], 0, 0
            while pos < n:
                m = min(int(sizes[k % len(sizes)]), n - pos)
                p, q = ch.feed(iq[pos:pos + m])
                outs.append(p)
                qs.append(q)
                pos += m
                k += 1
            assert np.array_equal(np.concatenate(outs), whole[c])
            assert np.array_equal(np.concatenate(qs), whole_q[c])
            ch.close()


@pytest.mark.parametrize("buf,decim,ntaps", [(4096, 96, 128), (16384, 96, 128), (1000, 96, 128), (4000, 25, 128),
                                              (4096, 40, 128), (4096, 400, 512)])
def test_twoslot_walk_equals_closed_form(pkg, ora, buf, decim, ntaps):
    """The reference's sb_active/sb_next walk (direct_fir.c:328-417) restated buffer by buffer gives the
    closed-form stream for uniform buffers."""
    fs = 2400000
    offs = [101000]
    cre, cim, incr = _plan(pkg, ora, offs, fs=fs, decim=decim, ntaps=ntaps)
    nb = 40
    iq = pkg.synth.random_iq(buf * nb, seed=buf + decim, full_scale=False)
    p2, q2 = ora.twoslot_run(iq, buf, cre[0], cim[0], decim, incr[0])
    p1, q1 = ora.run_channels(iq, cre, cim, incr, decim, want_iq=True)
    assert len(p2) == p1.shape[1]
    assert np.array_equal(p2, p1[0]) and np.array_equal(q2, q1[0])


def test_path_regression_vector(pkg, ora, golden_dir):
    g = np.load(os.path.join(golden_dir, "path_oracle.npz"))
    fs, decim = int(g["fs"]), int(g["decim"])
    for c, (o, gain) in enumerate(zip(g["offsets"], g["gains"])):
        cre, cim = ora.make_taps(g["lpf"], int(o), fs, float(gain))
        assert np.array_equal(cre, g["cre"][c]) and np.array_equal(cim, g["cim"][c])
        assert np.array_equal(ora.rot_incr(int(o), fs, decim), g["incr"][c])
    pcm, q = ora.run_channels(g["iq"], g["cre"], g["cim"], g["incr"], decim, want_iq=True)
    assert np.array_equal(pcm, g["pcm"]) and np.array_equal(q, g["filt_iq"])


def test_threaded_runner_matches_single_thread(pkg, ora):
    offs = list(pkg.synth.channel_offsets(12))
    cre, cim, incr = _plan(pkg, ora, offs)
    iq = pkg.synth.synth_iq(96 * 500 + 128, 2400000, offs[:3], seed=4)
    a, _ = ora.run_channels(iq, cre, cim, incr, 96, threads=1)
    b, _ = ora.run_channels(iq, cre, cim, incr, 96, threads=4)
    assert np.array_equal(a, b)


def test_int32_wraparound_is_exercised(pkg, ora):
    """Full-scale input with a large gain overflows the int32 accumulator; the oracle must wrap like
    the reference's plain int32 arithmetic on x86 (filter/complex.h:44-45)."""
    fs, decim = 2400000, 96
    taps = pkg.synth.design_lpf(128, 400000.0, fs)
    cre, cim = ora.make_taps(taps, 0, fs, 5.0)  # sum(c) = 5 * 16384, peak tap still an int16
    incr = ora.rot_incr(0, fs, decim)
    iq = np.full((4096, 2), 32767, np.int16)
    pcm, q = ora.run_channels(iq, cre[None], cim[None], incr[None], decim, want_iq=True)
    acc = int(np.sum(cre.astype(np.int64) * 32767 - cim.astype(np.int64) * 32767))
    assert abs(acc) > 2 ** 31  # really overflows
    wrapped = ((acc + 2 ** 31) % 2 ** 32) - 2 ** 31
    f = (wrapped >> 14) + ((wrapped >> 13) & 1)
    f16 = ((f + 2 ** 15) % 2 ** 16) - 2 ** 15
    # incr is exactly (16384, 0) so the derotator is the identity
    assert int(q[0, 0, 0]) == f16
