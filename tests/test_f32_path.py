"""BASELINE.json configs[4] ("fp32 vs int16 IQ path") / SURVEY.md 8d config 5: the channel path on float32 IQ.

The reference has no floating-point path, so the oracle is oracle/f32_oracle.c - the integer path's algorithm with
the Q14 quantisation steps removed, in fp64 (parity unpinned: there is nothing in the reference to pin it to).
Tolerances, from BASELINE.json north_star ("within 1 LSB (int16) / 1e-5 rel for the float FIR/atan2 stage"):

  * derotated filtered IQ, per sample: |gpu - oracle| <= 1e-5 * sum_i |c_i| |x_(nD+i)| - relative to the magnitude of what
    that output's window sums (the forward error bound of an fp32 dot product; relative to the OUTPUT a bound cannot hold
    where a channel's stop band cancels the sum to nothing), and, as before, <= 1e-5 of the run's largest magnitude; the
    share of samples that also meet 1e-5 relative to their own magnitude is reported by the test
  * float PCM (phi/pi*16384) of channels that carry a signal: circular difference <= 1e-5 * 16384 (0.16 LSB)
  * float PCM of every channel (noise-only ones included): <= 1 LSB, except where |s| is so small that the angle is
    ill-conditioned (|o[n]| |o[n-1]| below 1e-4 of the run's largest) - the share of samples that excludes is computed,
    bounded on the signal-carrying channels and printed (pytest -s)
"""
import numpy as np
import pytest


def _circ(d):
    return np.abs((d + 16384.0) % 32768.0 - 16384.0)


def test_oracle_f32_taps_and_int16_path_agree(pkg, ora):
    """The fp64 taps are the integer path's taps before truncation, and on a strong carrier the fp64 PCM and the
    reference-exact integer PCM differ by quantisation only (a couple of LSB)."""
    fs, decim = 2400000, 96
    lpf = pkg.synth.design_lpf(128, 12500.0, fs)
    off = -431250
    ch = ora.F32Channel(off, fs, decim, lpf, 1.0)
    re, im = ch.taps()
    cre, cim = ora.make_taps(lpf, off, fs, 1.0)
    assert np.array_equal(np.trunc(re * 16384.0).astype(np.int16), cre)
    assert np.array_equal(np.trunc(im * 16384.0).astype(np.int16), cim)
    iq = pkg.synth.synth_iq(1 << 16, fs, [off], seed=3)
    pcm, _ = ch.push(iq.astype(np.float32))
    ire, iim = ora.rot_incr(off, fs, decim)
    pcm_i, _ = ora.run_channels(iq, cre[None, :], cim[None, :], np.array([[ire, iim]], np.int16), decim)
    n = min(len(pcm), pcm_i.shape[1])
    assert n == (len(iq) - 128) // decim + 1
    # the first outputs differ by construction (the recursive Q14 rotator has not decayed yet / history 0)
    d = _circ(pcm[8:n] - pcm_i[0, 8:n].astype(np.float64))
    assert np.percentile(d, 99) < 6.0 and d.max() < 40.0
    # chunking independence of the oracle itself
    ch2 = ora.F32Channel(off, fs, decim, lpf, 1.0)
    x = iq.astype(np.float32)
    parts = [ch2.push(x[i:i + 5003])[0] for i in range(0, len(x), 5003)]
    assert np.allclose(np.concatenate(parts), pcm, rtol=0, atol=1e-9)


def _run_case(pkg, ora, fs, decim, ntaps, nch, nsamp, chunks, seed, cutoff=12500.0, scale=1.0, packed_fma=False,
              tile_kernel=False, want_iq=True):
    lpf = pkg.synth.design_lpf(ntaps, cutoff, fs)
    offs = pkg.synth.channel_offsets(nch, fs)
    rng = np.random.RandomState(seed)
    gains = 0.5 + rng.rand(nch)
    active = list(range(0, nch, max(1, nch // 6)))[:6]
    iq = pkg.synth.synth_iq(nsamp, fs, [offs[a] for a in active], seed=seed).astype(np.float32) * np.float32(scale)
    eng = pkg.F32Engine(fs, decim, max(chunks), device=0, want_iq=want_iq, packed_fma=packed_fma, tile_kernel=tile_kernel)
    for o, g in zip(offs, gains):
        eng.add_channel(int(o), lpf, float(g))
    eng.commit()
    got_f, got_i, got_q = [], [], []
    pos, k = 0, 0
    while pos < nsamp:
        n = min(chunks[k % len(chunks)], nsamp - pos)
        pf, pi, q = eng.process_host(iq[pos:pos + n])
        got_f.append(pf)
        got_i.append(pi)
        got_q.append(q if want_iq else np.zeros(pf.shape + (2,), np.float32))
        pos += n
        k += 1
    eng.close()
    gf, gi, gq = np.concatenate(got_f, 1), np.concatenate(got_i, 1), np.concatenate(got_q, 1)
    nout = (nsamp - ntaps) // decim + 1 if nsamp >= ntaps else 0
    assert gf.shape[1] == nout
    ref_p, ref_q = np.zeros((nch, nout)), np.zeros((nch, nout, 2))
    for c in range(nch):
        ch = ora.F32Channel(int(offs[c]), fs, decim, lpf, float(gains[c]))
        p, q = ch.push(iq)
        ref_p[c], ref_q[c] = p, q
        ch.close()
    full = np.abs(ref_q).max()
    # FIR + derotation
    assert not want_iq or np.abs(gq - ref_q).max() <= 1e-5 * full
    rel_own = None
    if want_iq:
        # per sample: against what the output's window sums in magnitude (|taps| . |samples| of that window)
        xm = np.hypot(iq[:, 0].astype(np.float64), iq[:, 1].astype(np.float64))
        win = np.lib.stride_tricks.sliding_window_view(xm, ntaps)[::decim][:nout]
        err = np.hypot(gq[:, :, 0] - ref_q[:, :, 0], gq[:, :, 1] - ref_q[:, :, 1])
        for c in range(nch):
            ch = ora.F32Channel(int(offs[c]), fs, decim, lpf, float(gains[c]))
            tre, tim = ch.taps()
            ch.close()
            bound = 1e-5 * (win @ np.hypot(tre, tim))
            worst = np.argmax(err[c] - bound)
            assert err[c][worst] <= bound[worst], (c, worst, err[c][worst], bound[worst])
        own = np.hypot(ref_q[:, :, 0], ref_q[:, :, 1])
        rel_own = float((err <= 1e-5 * np.maximum(own, 1e-300)).mean())
    # discriminator
    d = _circ(gf.astype(np.float64) - ref_p)
    assert d[active].max() <= 1e-5 * 16384.0
    mag = np.hypot(ref_q[:, :, 0], ref_q[:, :, 1])
    cond = mag[:, 1:] * mag[:, :-1] > 1e-4 * full * full
    assert d[:, 1:][cond].max() <= 1.0
    # how much the conditioning mask leaves out: nothing on the channels that carry a signal, and it is said for the rest
    excluded_all = 1.0 - float(cond.mean())
    excluded_active = 1.0 - float(cond[active].mean())
    assert excluded_active <= 0.01, excluded_active
    print(f"f32 path {fs} Hz / {decim}, {ntaps} taps, {nch} channels: 1-LSB check excludes {100 * excluded_all:.1f} % of all samples as "
          f"ill-conditioned ({100 * excluded_active:.2f} % on the {len(active)} signal-carrying channels)"
          + (f"; {100 * rel_own:.1f} % of the filtered-IQ samples within 1e-5 of their OWN magnitude" if rel_own is not None else ""))
    # int16 PCM is the float PCM truncated (multifm/fm_demod.c:72)
    assert np.array_equal(gi, np.trunc(gf).astype(np.int16))
    return d


@pytest.mark.gpu
@pytest.mark.parametrize("fs,decim,ntaps,nch,chunks", [
    (2400000, 96, 128, 64, [1 << 18]),                 # configs[1] shape
    (2400000, 96, 128, 5, [4096, 100, 50000, 37, 1]),  # ragged blocks, blocks shorter than the filter
    (10000000, 400, 512, 20, [1 << 17]),               # configs[4] shape (tap chunks of 128)
    (1200000, 25, 127, 3, [30001]),                    # odd decimation, odd filter length
    (1000000, 40, 40, 70, [65536]),                    # taps == decimation, two channel groups
])
def test_gpu_f32_path_matches_fp64_oracle(pkg, ora, fs, decim, ntaps, nch, chunks):
    _run_case(pkg, ora, fs, decim, ntaps, nch, 300000, chunks, seed=11)


@pytest.mark.gpu
def test_gpu_f32_path_without_debug_iq_and_round1_kernel(pkg, ora):
    """Without MFM_F32_WANT_IQ the persistent kernel stores through its plain path (whole tiles, no per-column
    predicates) - the way bench.py runs it; MFM_F32_TILE_KERNEL selects the round-1 kernel (one workgroup per tile)."""
    _run_case(pkg, ora, 2400000, 96, 128, 64, 400000, [1 << 18, 9001], seed=12, want_iq=False)
    _run_case(pkg, ora, 2400000, 96, 128, 70, 300000, [1 << 17], seed=13, want_iq=False)    # a partly filled channel group
    _run_case(pkg, ora, 10000000, 400, 512, 20, 300000, [1 << 17], seed=14, want_iq=False)
    _run_case(pkg, ora, 2400000, 96, 128, 64, 300000, [1 << 18], seed=11, tile_kernel=True)
    _run_case(pkg, ora, 1200000, 25, 127, 3, 300000, [30001], seed=11, tile_kernel=True, want_iq=False)


@pytest.mark.gpu
def test_gpu_f32_packed_fma_variant_matches_too(pkg, ora):
    """MFM_F32_PACKED_FMA selects the v_pk_fma_f32 form of the multiply (kept for A/B runs): same tolerances."""
    _run_case(pkg, ora, 2400000, 96, 128, 20, 300000, [70000, 4099], seed=4, packed_fma=True)
    _run_case(pkg, ora, 10000000, 400, 512, 9, 300000, [1 << 17], seed=6, packed_fma=True)


@pytest.mark.gpu
def test_gpu_f32_path_is_scale_invariant_and_restartable(pkg, ora):
    """atan2 does not care about the input scale: normalised (+-1) float IQ gives the same PCM within tolerance;
    a second engine restarts the stream from zero history."""
    d1 = _run_case(pkg, ora, 2400000, 96, 128, 8, 200000, [65536], seed=5, scale=1.0)
    d2 = _run_case(pkg, ora, 2400000, 96, 128, 8, 200000, [65536], seed=5, scale=1.0 / 32768.0)
    assert d1.shape == d2.shape


@pytest.mark.gpu
def test_gpu_f32_full_size_block_is_chunking_independent(pkg):
    """2^24 samples x 64 channels (the bench's float block) in one call and in ragged pieces: the stream position,
    the carried discriminator history and the derotation phase must line up, so the two runs agree to rounding
    (a tile boundary moves with the chunking, and with it the split of w^n into w^(tile start) * w^lane)."""
    fs, decim, taps, offs, gains = pkg.synth.plan("cfg2_64ch", nr_channels=64)
    n = 1 << 24
    base = pkg.synth.synth_iq(1 << 20, fs, offs[::8][:8], seed=7).astype(np.float32)
    iq = np.tile(base, (n // base.shape[0], 1))
    outs = []
    for chunks in ([n], [5000001, 77, 3000000, n]):
        eng = pkg.F32Engine(fs, decim, n, device=0, want_iq=True)
        for o, g in zip(offs, gains):
            eng.add_channel(int(o), taps, float(g))
        eng.commit()
        pf, pq, pos, k = [], [], 0, 0
        while pos < n:
            m = min(chunks[k], n - pos)
            f, _, q = eng.process_host(iq[pos:pos + m])
            pf.append(f)
            pq.append(q)
            pos += m
            k += 1
        eng.close()
        outs.append((np.concatenate(pf, 1), np.concatenate(pq, 1)))
    (f0, q0), (f1, q1) = outs
    assert f0.shape == f1.shape == (64, (n - len(taps)) // decim + 1)
    full = np.abs(q0).max()
    assert np.abs(q0 - q1).max() <= 2e-6 * full
    mag = np.hypot(q0[:, :, 0], q0[:, :, 1])
    ok = mag[:, 1:] * mag[:, :-1] > 1e-4 * full * full
    assert _circ(f0.astype(np.float64) - f1)[:, 1:][ok].max() <= 0.05


@pytest.mark.gpu
def test_gpu_f32_pcm_feeds_the_integer_stages(pkg, ora):
    """The int16 PCM of the float path has the integer engine's layout: the resampler takes it in place."""
    import ctypes as C
    fs, decim, nch = 1152000, 24, 4
    lpf = pkg.synth.design_lpf(128, 12500.0, fs)
    offs = pkg.synth.channel_offsets(nch, fs)
    iq = pkg.synth.synth_iq(1 << 18, fs, offs[:2], seed=2).astype(np.float32)
    eng = pkg.F32Engine(fs, decim, 1 << 18, device=0)
    for o in offs:
        eng.add_channel(int(o), lpf, 1.0)
    eng.commit()
    import torch
    d_iq = torch.from_numpy(iq.reshape(-1)).cuda()
    blk = eng.process_device(d_iq.data_ptr(), iq.shape[0])
    rtaps = ora.quantize_taps(pkg.synth.design_lpf(81, 0.45 / 5, 1.0) * 4)
    rs = pkg.Resampler(nch, rtaps, 4, 5, eng.max_out(), device=0)
    yptr, ystride, ny = rs.process_device(blk.d_pcm_i16, blk.stride, blk.nr_out)
    torch.cuda.synchronize()
    pcm = np.zeros((nch, blk.stride), np.int16)
    lib = pkg.load_library()
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    assert hip.hipMemcpy(pcm.ctypes.data, blk.d_pcm_i16, pcm.nbytes, 2) == 0
    y = np.zeros((nch, ystride), np.int16)
    assert hip.hipMemcpy(y.ctypes.data, yptr, y.nbytes, 2) == 0
    for c in range(nch):
        o = ora.Resampler(rtaps, 4, 5)
        assert np.array_equal(o.feed(pcm[c, :blk.nr_out]), y[c, :ny])
    rs.close()
    eng.close()
    del lib
