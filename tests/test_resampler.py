"""the PCM rational resampler + DC blocker (filter/polyphase_fir.c, filter/utils.c,
filter/dc_blocker.h) - oracle self-checks on the CPU, bit-exact GPU parity through the C ABI, and the
device-resident chain engine -> resampler of BASELINE configs[3] (48 kS/s PCM -> 38 400 Hz)."""
import ctypes as C

import numpy as np
import pytest


def _design(pkg, ntaps, interp, decim):
    # low-pass at min(1/I, 1/D) of the interpolated rate, gain I (the usual rational-resampler design)
    fc = 0.45 / max(interp, decim)
    return pkg.synth.design_lpf(ntaps, fc, 1.0) * interp


def test_oracle_resampler_known_structure(pkg, ora):
    """Phase scatter and the strict '>' rule of polyphase_fir.c:70-83,184, checked against a direct numpy
    evaluation of the same definition."""
    interp, decim = 4, 5
    taps = ora.quantize_taps(_design(pkg, 41, interp, decim))
    assert taps[20] == int(_design(pkg, 41, interp, decim)[20] * 16384)  # truncation, decoder.c:530-533
    rs = ora.Resampler(taps, interp, decim)
    plen = rs.phase_len()
    assert plen == 12  # ceil(41/4) = 11 -> rounded up to a multiple of 4
    rng = np.random.RandomState(1)
    x = rng.randint(-20000, 20000, size=3000).astype(np.int16)
    y = rs.feed(x)
    ph = np.zeros((interp, plen), np.int64)
    for i, c in enumerate(taps):
        ph[i % interp, i // interp] = c
    out, p, pos = [], 0, 0
    while len(x) - pos > plen:
        acc = int(np.dot(ph[p], x[pos:pos + plen].astype(np.int64)))
        acc = ((acc + 2 ** 31) % 2 ** 32) - 2 ** 31
        v = (acc >> 14) + ((acc >> 13) & 1)
        out.append(((v + 2 ** 15) % 2 ** 16) - 2 ** 15)
        p += decim
        pos += p // interp
        p %= interp
    assert np.array_equal(y, np.array(out, np.int16))
    rs2 = ora.Resampler(taps, interp, decim)
    parts = [rs2.feed(x[i:i + 257]) for i in range(0, len(x), 257)]
    assert np.array_equal(np.concatenate(parts), y)  # chunking independence


def test_oracle_dc_blocker_removes_dc(ora):
    rs = ora.Resampler(np.array([16384], np.int16), 1, 1, dc_pole=0.999)
    x = (np.full(20000, 3000) + 1000 * np.sin(np.arange(20000) * 0.3)).astype(np.int16)
    y = rs.feed(x)
    assert abs(int(y[-5000:].astype(np.int64).mean())) < 40 and y[-5000:].std() > 500


@pytest.mark.gpu
@pytest.mark.parametrize("interp,decim,ntaps,nch", [(4, 5, 81, 5), (16, 25, 821, 3), (3, 2, 41, 2), (1, 1, 9, 1),
                                                    (1, 4, 33, 7), (7, 3, 50, 2)])
def test_gpu_resampler_matches_oracle(pkg, ora, interp, decim, ntaps, nch):
    """48k -> 38.4k for POCSAG (4/5), 25k -> 16k for FLEX (16/25, 821 taps like etc/resampler_filter.json), the
    3/2 case of the reference's own polyphase smoke test, plus pure decimation / interpolation-heavy shapes;
    ragged chunks; with and without inversion and DC blocking."""
    taps = ora.quantize_taps(_design(pkg, ntaps, interp, decim))
    rng = np.random.RandomState(interp * 100 + decim)
    n = 30000
    x = rng.randint(-32768, 32768, size=(nch, n)).astype(np.int16)
    for dc_pole, invert in ((None, False), (0.9999, True)):
        gpu = pkg.Resampler(nch, taps, interp, decim, 8192, device=0, invert=invert, dc_pole=dc_pole)
        refs = [ora.Resampler(taps, interp, decim, dc_pole=dc_pole, invert=invert) for _ in range(nch)]
        sizes = [1024, 1, 7, 4096, 100, 8192, 2048]
        pos, k, got, want = 0, 0, [], [[] for _ in range(nch)]
        while pos < n:
            m = min(sizes[k % len(sizes)], n - pos)
            got.append(gpu.process_host(x[:, pos:pos + m]))
            for c in range(nch):
                want[c].append(refs[c].feed(x[c, pos:pos + m]))
            pos += m
            k += 1
        got = np.concatenate(got, axis=1)
        want = np.stack([np.concatenate(w) for w in want])
        gpu.close()
        assert got.shape == want.shape, (got.shape, want.shape)
        assert np.array_equal(got, want), f"{int((got != want).sum())} samples differ (dc={dc_pole}, invert={invert})"


@pytest.mark.gpu
@pytest.mark.parametrize("interp,decim,ntaps", [(16, 25, 821), (4, 5, 81), (16, 25, 400), (8, 5, 200), (2, 3, 96), (1, 2, 120)])
def test_gpu_resampler_matrix_form_wraps_like_the_reference(pkg, ora, interp, decim, ntaps):
    """The matrix-core form (16 D / I integer) splits taps and samples into bytes; full-scale taps up to +-32639 and
    full-scale samples make the int32 sums wrap (filter/utils.c:94-103) - it must wrap identically, block after block,
    and give the same bits as the v_dot2 form (MFM_RS_FORCE_DOT2).  One tap beyond +-32639 (32767): the engine must
    fall back to v_dot2 by itself."""
    rng = np.random.RandomState(ntaps)
    nch, n = 4, 150000
    x = rng.randint(-32768, 32768, size=(nch, n)).astype(np.int16)
    x[1] = 32767
    x[2] = -32768
    for big in (32639, 32767):
        taps = rng.randint(-32639, 32640, size=ntaps).astype(np.int16)
        taps[rng.randint(ntaps)] = big
        taps[rng.randint(ntaps)] = -32639
        refs = [ora.Resampler(taps, interp, decim) for _ in range(nch)]
        want = np.stack([r.feed(x[c]) for c, r in enumerate(refs)])
        for force in (False, True):
            gpu = pkg.Resampler(nch, taps, interp, decim, 65536, device=0, force_dot2=force)
            got, pos = [], 0
            for m in (65536, 3, 40001, 65536):
                got.append(gpu.process_host(x[:, pos:pos + m]))
                pos += m
            gpu.close()
            got = np.concatenate(got, axis=1)
            assert got.shape == want[:, :got.shape[1]].shape and got.shape[1] > 0
            assert np.array_equal(got, want[:, :got.shape[1]]), f"big={big} force_dot2={force}"


@pytest.mark.gpu
def test_gpu_chain_engine_to_resampler_stays_on_device(pkg, ora):
    """BASELINE configs[3] front half: etc/pocsag_rtlsdr.json values (fs 1.2 MS/s, D 25 -> 48 kS/s PCM, channel 0
    with dBGain 4.0), then 4/5 to the 38 400 Hz the POCSAG decoder wants; PCM goes from the channel kernel to
    the resampler without leaving HBM."""
    fs, decim, taps, offs, gains = pkg.synth.plan("pocsag_rtlsdr")
    blk = 1 << 18
    iq = pkg.synth.synth_iq(blk * 3, fs, offs, seed=12)
    eng = pkg.Engine(fs, decim, blk, device=0, flags=pkg.binding.MFM_F_DEVICE_ONLY)
    for o, g in zip(offs, gains):
        eng.add_channel(int(o), taps, float(g))
    eng.commit()
    rtaps = ora.quantize_taps(_design(pkg, 81, 4, 5))
    max_pcm = blk // decim + 8
    rs = pkg.Resampler(len(offs), rtaps, 4, 5, max_pcm, device=0)
    rt = C.CDLL("libamdhip64.so")
    rt.hipMemcpy2D.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int]
    outs = []
    for b in range(3):
        assert eng.push(iq[b * blk:(b + 1) * blk]) == 0
        dptr, stride, nout, _ = eng.last_output_device()
        yptr, ystride, ny = rs.process_device(dptr, stride, nout, stream=eng.stream)
        eng.sync()
        host = np.zeros((len(offs), max(ny, 1)), np.int16)
        if ny:
            assert rt.hipMemcpy2D(host.ctypes.data, host.shape[1] * 2, yptr, ystride * 2, ny * 2, len(offs), 2) == 0
        outs.append(host[:, :ny])
    got = np.concatenate(outs, axis=1)
    eng.close()
    rs.close()
    cre = np.stack([ora.make_taps(taps, int(o), fs, float(g))[0] for o, g in zip(offs, gains)])
    cim = np.stack([ora.make_taps(taps, int(o), fs, float(g))[1] for o, g in zip(offs, gains)])
    incr = np.stack([ora.rot_incr(int(o), fs, decim) for o in offs])
    pcm, _ = ora.run_channels(iq, cre, cim, incr, decim)
    want = np.stack([ora.Resampler(rtaps, 4, 5).feed(pcm[c]) for c in range(len(offs))])
    assert got.shape == want.shape and np.array_equal(got, want)
