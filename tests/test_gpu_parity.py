"""Parity tests proper: the HIP path, called through the C ABI, against the oracle on the same seeded
input.  Bar: bit-exact PCM and filtered IQ (the float stage is reproduced exactly, see
tests/test_numerics_host.py, so no tolerance is needed; north_star would allow 1 LSB)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


# "auto" = the matrix-core kernels where they apply (second generation when decimation % 8 == 0 and <= 128 taps, and - round 5 -
# for filters of 129..512 taps at any decimation without a filtered-IQ consumer; else the first), "mfma1" = first-generation
# matrix kernel forced, "dot2" = forced v_dot2 kernel.  All three give the same bits.
KERNELS = ["auto", "mfma1", "dot2"]


def _mk_engine(pkg, fs, decim, taps, offs, gains=None, max_block=1 << 16, want_iq=False, flags=0, kernel="auto"):
    if kernel == "dot2":
        flags |= pkg.binding.MFM_F_FORCE_DOT2
    if kernel == "mfma1":
        flags |= pkg.binding.MFM_F_FORCE_MFMA_V1
    if kernel == "mfma1s":  # filters of 129..512 taps: the streamed-taps form instead of the resident one
        flags |= pkg.binding.MFM_F_STREAM_TAPS
    eng = pkg.Engine(fs, decim, max_block, device=0, flags=flags)
    gains = gains if gains is not None else [1.0] * len(offs)
    for o, g in zip(offs, gains):
        eng.add_channel(int(o), taps, float(g), want_iq=want_iq)
    eng.commit()
    variant = eng.stats()["kernel_variant"]
    if kernel == "auto" and decim == 25 and len(taps) <= 150 and np.abs(_all_taps(eng, len(offs))).max() <= 32639:
        assert variant == 2, "decimation 25, with or without filtered IQ: the second-generation kernel on padded rows applies"
    if kernel == "dot2":
        assert variant == 0
    elif decim % 8 == 0 and len(taps) <= 128 and np.abs(_all_taps(eng, len(offs))).max() <= 32639:
        if kernel == "mfma1":
            assert variant == 1, "the first-generation matrix-core kernel should have been selected"
        else:
            assert variant in (1, 2), "a matrix-core kernel should have been selected"
            if decim % 8 == 0 and decim <= 128:
                assert variant == 2, "decimation % 8 == 0, <= 128 taps: the second-generation kernel applies"
    return eng


def _all_taps(eng, nch):
    return np.stack([np.stack(eng.get_channel(c)[:2]) for c in range(nch)])


def _oracle_tables(eng, nch):
    cre = np.stack([eng.get_channel(c)[0] for c in range(nch)])
    cim = np.stack([eng.get_channel(c)[1] for c in range(nch)])
    incr = np.stack([eng.get_channel(c)[2] for c in range(nch)])
    return cre, cim, incr


def _check(pkg, ora, fs, decim, taps, offs, iq, block, gains=None, want_iq=True, threads=8, kernel="auto"):
    eng = _mk_engine(pkg, fs, decim, taps, offs, gains, max_block=max(block, 1), want_iq=want_iq, kernel=kernel)
    cre, cim, incr = _oracle_tables(eng, len(offs))
    pcm, q = eng.run(iq, block)
    ref, refq = ora.run_channels(iq, cre, cim, incr, decim, threads=threads, want_iq=want_iq)
    eng.close()
    assert pcm.shape == ref.shape, (pcm.shape, ref.shape)
    if not np.array_equal(pcm, ref):
        bad = np.argwhere(pcm != ref)
        raise AssertionError(f"{len(bad)} PCM samples differ; first at (chan, n) = {bad[0]}: "
                             f"hip {pcm[tuple(bad[0])]} oracle {ref[tuple(bad[0])]}")
    if want_iq:
        assert q is not None and np.array_equal(q, refq), "filtered IQ stream differs"
    return pcm


def test_golden_path_vector(pkg, ora, golden_dir):
    g = np.load(os.path.join(golden_dir, "path_oracle.npz"))
    fs, decim = int(g["fs"]), int(g["decim"])
    eng = pkg.Engine(fs, decim, 1 << 15, device=0)
    for c in range(len(g["offsets"])):
        eng.add_channel_q14(g["cre"][c], g["cim"][c], g["incr"][c], want_iq=True)
    eng.commit()
    pcm, q = eng.run(g["iq"], 1 << 15)
    eng.close()
    assert np.array_equal(pcm, g["pcm"]) and np.array_equal(q, g["filt_iq"])


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("name", ["multifm_1ch", "multifm_1ch_2400k", "pocsag_rtlsdr"])
def test_reference_shaped_configs(pkg, ora, name, kernel):
    """BASELINE configs[0] and [3]: etc/multifm_1ch.json values (fs 1.0 MS/s, D 40) and its 2.4 MS/s / D 96
    variant, etc/pocsag_rtlsdr.json (fs 1.2 MS/s, D 25, dBGain 4.0 on channel 0), file_if-sized 4096-sample
    buffers (multifm/file_if.c:18)."""
    fs, decim, taps, offs, gains = pkg.synth.plan(name)
    iq = pkg.synth.synth_iq(4096 * 60, fs, offs, seed=21)
    _check(pkg, ora, fs, decim, taps, offs, iq, 4096, gains=gains, kernel=kernel)
    if name == "pocsag_rtlsdr" and kernel != "dot2":
        # decimation 25 is not a multiple of 8: LDS rows padded from 50 to 64 bytes, zero taps over the padding
        # (filter/direct_fir.c:328-417 has no restriction on the decimation; neither have the matrix kernels): the second
        # generation on its padded-row layout since round 4, the first generation when forced
        for want_iq in (False, True):   # (True: the DBG_IQ instances of the padded-row layout, round 6)
            eng = _mk_engine(pkg, fs, decim, taps, offs, gains, max_block=4096, kernel=kernel, want_iq=want_iq)
            st = eng.stats()
            eng.close()
            assert st["kernel_variant"] == (2 if kernel == "auto" else 1)
    if name == "multifm_1ch" and kernel == "auto":
        # etc/multifm.json, etc/multifm_1ch.json: decimation 40 - a multiple of 8, not of 32: the second-generation
        # kernel on its chunk-row LDS layout
        eng = _mk_engine(pkg, fs, decim, taps, offs, gains, max_block=4096, kernel=kernel)
        st = eng.stats()
        eng.close()
        assert st["kernel_variant"] == 2


# every channelizer geometry the reference ships under etc/: (name, fs, centre, decimation, channel centre frequencies, gains in
# dB, the low-pass that goes with it by sample rate: its length and cut-off, the front end's sample format and buffer size)
ETC_CONFIGS = [
    # etc/multifm.json (8 channels; etc/multifm_1ch.json is its first) + etc/flex_25khz_lpf.json (128 taps); RTL-SDR bytes,
    # 131 072-sample buffers (multifm/rtl_sdr_if.c:46)
    ("multifm", 1000000, 929500000, 40, [929838000, 929538000, 929388000, 929938000, 929362000, 929662500, 929638000, 929612000],
     None, 128, 12500.0, "rtlsdr_u8", 131072),
    # etc/multifm_airspy.json, etc/multifm_usrp.json + etc/flex_25khz_lpf_3mhz.json (512 taps); int16, 16 384-sample buffers
    ("multifm_airspy", 3000000, 930500000, 120, [929612500], None, 512, 12500.0, "cs16", 16384),
    # etc/pocsag_rtlsdr.json + etc/pocsag_1200khz_fs.json (256 taps, hamming, 9 kHz); channel 1's gain key is misspelt
    # ("dbGain") in the file, so it runs at 0 dB (multifm/receiver.c reads "dBGain")
    ("pocsag_rtlsdr", 1200000, 152500000, 25, [152180000, 152008000], [4.0, 0.0], 256, 9000.0, "rtlsdr_u8", 131072),
    # etc/pocsag_airspy.json + etc/pocsag_narrow.json (256 taps, hamming, 4.8 kHz); int16, 262 144-sample buffers
    # (multifm/airspy_if.c:244-245)
    ("pocsag_airspy", 2500000, 152500000, 100, [152180000], [2.0], 256, 4800.0, "cs16", 262144),
    # etc/multifm_file.json: a cs8 capture at 8 738 133 S/s channelised WITHOUT decimation, 4096-sample buffers
    # (multifm/file_if.c:18); no filter file goes with it by name: 128 taps
    ("multifm_file", 8738133, 1692000000, 1, [1691000000], None, 128, 400000.0, "cs8", 4096),
]


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("cfg", ETC_CONFIGS, ids=[c[0] for c in ETC_CONFIGS])
def test_every_configuration_under_etc(pkg, ora, cfg, kernel):
    """The reference's own configurations - sample rate, decimation, channel centres and gains as the files under etc/ have
    them, filters of the lengths its filter files have, the sample format and buffer size of the front end each is written
    for (8-bit sources pushed as bytes) - against the oracle on the widened input, every kernel."""
    name, fs, centre, decim, chans, gains_db, ntaps, cutoff, fmt_name, buf = cfg
    b = pkg.binding
    offs = np.array([f - centre for f in chans], dtype=np.int64)
    gains = [10.0 ** (g / 10.0) for g in gains_db] if gains_db else [1.0] * len(offs)   # multifm/receiver.c: dBGain -> linear
    taps = pkg.synth.design_lpf(ntaps, cutoff, fs)
    n = buf * 5 + 1234 if decim > 1 else buf * 9 + 77
    fmt = {"cs16": b.MFM_IN_CS16, "cs8": b.MFM_IN_CS8, "rtlsdr_u8": b.MFM_IN_RTLSDR_U8}[fmt_name]
    sig = pkg.synth.synth_iq(n, fs, offs, seed=len(name))
    if fmt == b.MFM_IN_CS16:
        raw, iq = sig, sig
    elif fmt == b.MFM_IN_CS8:
        raw = (sig >> 8).astype(np.int8).view(np.uint8)
        iq = ora.unpack_bytes(raw, fmt).reshape(-1, 2)
    else:
        raw = np.clip((sig.astype(np.int32) >> 7) + 127, 0, 255).astype(np.uint8)
        iq = ora.unpack_bytes(raw, fmt).reshape(-1, 2)
    eng = _mk_engine(pkg, fs, decim, taps, offs, gains, max_block=buf, kernel=kernel)
    cre, cim, incr = _oracle_tables(eng, len(offs))
    parts = []
    for lo in range(0, n, buf):
        blk = raw[lo:lo + buf]
        while True:
            rc = eng.push(blk.reshape(-1)) if fmt == b.MFM_IN_CS16 else eng.push_bytes(blk, fmt)
            if rc == 0:
                break
            assert rc == b.MFM_E_BUSY, eng.lib.mfm_last_error()
            parts.append(eng.fetch()[1])
    eng.sync()
    while True:
        got = eng.fetch()
        if got is None:
            break
        parts.append(got[1])
    eng.close()
    pcm = np.concatenate(parts, axis=1)
    ref, _ = ora.run_channels(iq, cre, cim, incr, decim, threads=8)
    assert pcm.shape == ref.shape, (pcm.shape, ref.shape)
    assert np.array_equal(pcm, ref), f"{name}: {np.count_nonzero(pcm != ref)} PCM samples differ"


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("decim,ntaps", [(1, 1), (1, 128), (2, 2), (2, 129), (3, 64), (4, 128), (5, 5), (5, 200), (6, 512), (7, 100)])
def test_decimations_below_eight(pkg, ora, decim, ntaps, kernel):
    """filter/direct_fir.c:328-417 decimates by any factor >= 1 (etc/multifm_file.json: 1); filters from one tap per output
    sample to 512 taps, ragged blocks, full-scale input, filtered IQ compared as well."""
    fs = 1000000
    taps = pkg.synth.design_lpf(ntaps, 100000.0, fs) if ntaps > 1 else np.array([0.9])
    offs = np.array([0, 250000, -123456, 31250, fs // (4 * decim) if decim > 1 else 7, -499999], dtype=np.int64)
    iq = pkg.synth.random_iq(20000 * decim + ntaps + 3, seed=decim * 1000 + ntaps)
    _check(pkg, ora, fs, decim, taps, offs, iq, 7001, kernel=kernel)


@pytest.mark.parametrize("decim,ntaps,nch", [(1, 128, 3), (1, 128, 70), (1, 16, 5), (1, 200, 9), (1, 512, 4), (2, 128, 64), (2, 33, 3),
                                             (2, 400, 7), (4, 128, 10), (4, 64, 130), (4, 512, 3), (1, 1, 2), (4, 4, 3)])
def test_decimations_1_2_4_run_on_the_matrix_cores(pkg, ora, decim, ntaps, nch):
    """etc/multifm_file.json channelises a cs8 capture WITHOUT decimating (filter/direct_fir.c:328-417 takes any factor >= 1).  A row
    of 2 D plane bytes is shorter than a 16-byte fragment there, so the long-filter kernel keeps 8 / D shifted copies of the tile's
    image and every column reads the copy in which its window starts aligned (mfm_kernel_v3l.hip, SHIFT; round 5 - the v_dot2
    kernel before).  int16 blocks and the three 8-bit forms as bytes, ragged block lengths, mixed rotator classes, 4 .. 16
    k-steps; since round 6 also with a channel that wants its filtered IQ (signalDebugFile, multifm/demod.c:75-81: a run-time
    switch in that kernel's epilogue)."""
    fs = 1000000
    taps = pkg.synth.design_lpf(ntaps, 100000.0, fs) if ntaps > 1 else np.array([0.9])
    rng = np.random.RandomState(decim * 100 + ntaps)
    offs = np.where(rng.randint(0, 2, size=nch) == 0, (fs // (4 * decim)) * rng.randint(-3, 4, size=nch),
                    rng.randint(-450000, 450000, size=nch)).astype(np.int64)
    n = 20000 * decim + ntaps + 3
    iq = pkg.synth.random_iq(n, seed=decim * 1000 + ntaps, full_scale=(ntaps == 128))
    eng = _mk_engine(pkg, fs, decim, taps, offs, max_block=1 << 15, want_iq=False)
    st = eng.stats()
    eng.close()
    assert st["kernel_variant"] == 2 and st["taps_resident"] == 1, st
    _check(pkg, ora, fs, decim, taps, offs, iq, 1 << 15, want_iq=False)
    _check(pkg, ora, fs, decim, taps, offs, iq, 7001, want_iq=False)
    eng = _mk_engine(pkg, fs, decim, taps, offs[:2], max_block=1 << 15, want_iq=True)
    assert eng.stats()["kernel_variant"] == 2
    eng.close()
    _check(pkg, ora, fs, decim, taps, offs, iq[: 4000 * decim + ntaps], 7001, want_iq=True)
    for fmt in (1, 2, 3):
        raw = np.random.RandomState(nch + fmt).randint(0, 256, size=(n, 2)).astype(np.uint8)
        sizes = [32768, 4096, 2, 30000, 8, 12346]
        blocks, pos, k = [], 0, 0
        while pos < n:
            m = min(sizes[k % len(sizes)], n - pos)
            m -= (m & 1) if fmt == 2 and m > 1 else 0
            blocks.append((raw[pos:pos + m], fmt))
            pos += m
            k += 1
        got, want, st8 = _ingest_8bit(pkg, ora, fs, decim, taps, offs, blocks, 1 << 15)
        assert st8["kernel_variant"] == 2 and st8["launches_8bit"] > 0, st8
        assert got.shape == want.shape and np.array_equal(got, want), (fmt, int((got != want).sum()))


@pytest.mark.parametrize("decim,ntaps,nch,want_iq", [(40, 128, 64, False), (40, 128, 9, True), (8, 8, 3, True), (8, 60, 20, False),
                                                     (16, 128, 5, False), (24, 100, 70, False), (48, 128, 64, True),
                                                     (56, 56, 3, False), (72, 128, 130, False), (88, 100, 7, True),
                                                     (104, 128, 16, False), (120, 128, 64, False)])
def test_decimations_that_are_multiples_of_8_run_on_the_second_generation_kernel(pkg, ora, decim, ntaps, nch, want_iq):
    """filter/direct_fir.c:328-417 has no restriction on the decimation.  Multiples of 8 that are not multiples of 32 (40:
    the reference's own etc/multifm*.json) keep the LDS image as rows of 16-byte chunks, four outputs' worth per slot, with
    the first three rows of a slot repeated behind the slot before (mfm_kernel.h, mfm_launch_v3::layout): every geometry
    of that family the kernel is built for, channel counts that leave waves and slices partly empty, ragged blocks and
    blocks shorter than a tile, mixed rotator classes, filtered-IQ output on and off, 8-bit input as bytes."""
    fs = 1000000
    taps = pkg.synth.design_lpf(ntaps, 12500.0, fs)
    rng = np.random.RandomState(decim)
    offs = np.where(rng.randint(0, 2, size=nch) == 0, (fs // decim) * rng.randint(-12, 12, size=nch),
                    rng.randint(-450000, 450000, size=nch)).astype(np.int32)
    n = decim * 2100 + ntaps + 13
    iq = pkg.synth.random_iq(n, seed=decim + nch, full_scale=(decim % 16 == 8))
    eng = _mk_engine(pkg, fs, decim, taps, offs, max_block=1 << 16, want_iq=want_iq)
    st = eng.stats()
    eng.close()
    assert st["kernel_variant"] == 2, st
    _check(pkg, ora, fs, decim, taps, offs, iq, 1 << 16, want_iq=want_iq)
    _check(pkg, ora, fs, decim, taps, offs, iq, decim * 130 + 7, want_iq=False)
    if not want_iq:
        raw = np.random.RandomState(nch).randint(0, 256, size=(n, 2)).astype(np.uint8)
        cut = min(40000, n // 2)
        blocks = [(raw[:cut], 3), (raw[cut:cut + 2], 3)] + [(raw[p:p + 60001], 3) for p in range(cut + 2, n, 60001)]
        got, want, st8 = _ingest_8bit(pkg, ora, fs, decim, taps, offs, blocks, 1 << 16)
        assert st8["kernel_variant"] == 2 and st8["launches_8bit"] == st8["launches"] > 0
        assert got.shape == want.shape and np.array_equal(got, want)


@pytest.mark.parametrize("kernel", KERNELS)
def test_cfg2_64_channels(pkg, ora, kernel):
    """BASELINE configs[1]: 64 channels, 25 kHz LPF (128 taps), D=96, 2.4 MS/s, 1 GPU."""
    fs, decim, taps, offs, gains = pkg.synth.plan("cfg2_64ch")
    iq = pkg.synth.synth_iq((1 << 20) + 4321, fs, offs[::7], seed=22)
    pcm = _check(pkg, ora, fs, decim, taps, offs, iq, 1 << 18, want_iq=False, kernel=kernel)
    assert pcm.shape == (64, ora.expected_outputs(len(iq), 128, decim))


@pytest.mark.parametrize("kernel", KERNELS)
def test_cfg3_shard_of_1024_channels(pkg, ora, kernel):
    """BASELINE configs[2]: one GPU's 128-channel shard of the 1024-channel plan."""
    fs, decim, taps, offs, gains = pkg.synth.plan("cfg3_1024ch")
    shard = offs[3 * 128:4 * 128]
    iq = pkg.synth.synth_iq(1 << 18, fs, shard[::16], seed=23)
    _check(pkg, ora, fs, decim, taps, shard, iq, 1 << 17, want_iq=False, kernel=kernel)


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("nch", [257, 512, 1024])
def test_many_channels_on_one_gpu(pkg, ora, nch, kernel):
    """north_star's target shape: >= 1024 channels on ONE GPU (the reference builds one demod_thread per channels[]
    entry without limit, multifm/receiver.c:195-244).  More than 256 channels = more than four 64-channel slices per
    tile and per-channel constants that no longer fit one LDS table; several blocks so that the carried sample and
    the rotator index of every channel cross a launch boundary; irregular block length."""
    fs, decim, taps, offs, gains = pkg.synth.plan("cfg3_1024ch", nr_channels=nch)
    iq = pkg.synth.synth_iq(3 * 50000 + 777, fs, offs[:: max(1, nch // 5)][:5], seed=nch)
    pcm = _check(pkg, ora, fs, decim, taps, offs, iq, 50000, gains=gains, want_iq=(nch == 257), threads=16,
                 kernel=kernel)
    assert pcm.shape[0] == nch


def test_2048_channels_at_the_airspy_geometry(pkg, ora):
    """BASELINE configs[4] whole channel set on one GPU: 2048 channels, fs 10 MS/s, D = 400, 512 taps (streamed taps,
    single-iteration tiles, 32 slices per tile)."""
    fs, decim, taps, offs, gains = pkg.synth.plan("cfg5_airspy", nr_channels=2048)
    iq = pkg.synth.synth_iq(2 * 60000 + 1234, fs, offs[::400][:5], seed=2048)
    eng = _mk_engine(pkg, fs, decim, taps, offs, gains, max_block=60000, want_iq=False)
    assert eng.stats()["kernel_variant"] == 2  # round 5: half-tile images of the long-filter kernel (decimation 400)
    eng.close()
    _check(pkg, ora, fs, decim, taps, offs, iq, 60000, gains=gains, want_iq=False, threads=16)
    _check(pkg, ora, fs, decim, taps, offs, iq, 60000, gains=gains, want_iq=False, threads=16, kernel="mfma1")


@pytest.mark.parametrize("ntaps,want_iq", [(512, False), (512, True), (256, False), (160, False)])
def test_long_filters_stream_their_taps_through_the_matrix_kernel(pkg, ora, ntaps, want_iq):
    """129..512 taps (the 512-tap etc/flex_25khz_lpf file of BASELINE configs[4]) at the 2.4 MS/s, D = 96 geometry: the
    MFMA kernel re-reads the tap fragments from L2 in chunks of 128 taps; 24 channels = three row blocks, several
    blocks so that tiles, passes and the carried state are all exercised."""
    fs, decim, _, offs, gains = pkg.synth.plan("cfg2_64ch", nr_channels=24)
    taps = pkg.synth.design_lpf(ntaps, 12500.0, fs)
    iq = pkg.synth.synth_iq(96 * 2500 + ntaps, fs, offs[:3], seed=ntaps)
    eng = _mk_engine(pkg, fs, decim, taps, offs, gains, max_block=1 << 16, want_iq=want_iq, kernel="mfma1s")
    assert eng.stats()["kernel_variant"] == 1 and eng.stats()["taps_resident"] == 0
    eng.close()
    _check(pkg, ora, fs, decim, taps, offs, iq, 1 << 16, gains=gains, want_iq=want_iq, kernel="mfma1s")
    # without the flag: the second generation's long-filter kernel - since round 6 also when a channel wants its filtered IQ
    # (signalDebugFile, multifm/demod.c:75-81: a run-time switch in that kernel's epilogue)
    eng = _mk_engine(pkg, fs, decim, taps, offs, gains, max_block=1 << 16, want_iq=want_iq)
    assert eng.stats()["kernel_variant"] == 2, eng.stats()
    eng.close()
    _check(pkg, ora, fs, decim, taps, offs, iq, 1 << 16, gains=gains, want_iq=want_iq)


@pytest.mark.parametrize("shape", ["lpf_div8", "lpf", "lpf_x8", "dense"])
@pytest.mark.parametrize("decim,ntaps", [(96, 512), (96, 256), (96, 300), (48, 129), (400, 512), (320, 512), (200, 256),
                                         (256, 256), (136, 160), (448, 512), (100, 400)])
def test_long_filters_keep_their_taps_in_registers(pkg, ora, decim, ntaps, shape):
    """Filters of 129..512 taps without a filtered-IQ consumer run the RESIDENT instances of the first-generation matrix
    kernel (all 8 / 16 k-steps of taps in registers, 256 registers, one workgroup per CU); MFM_F_STREAM_TAPS selects the
    round-1 form that re-reads them from L2.  Both against the oracle, over the instance families: two-iteration and
    single-iteration tiles, 2..8 staging chunks per thread, padded rows (D = 100), and the three tap-plane masks an
    instance can be built for - none (a low-pass whose taps all fit one byte, the configs[4] filter at 10 MS/s), a plain
    low-pass (high bytes in the middle k-steps only), the same at 18 dB (more of them), and dense random taps (every plane)."""
    fs = 4000000
    rng = np.random.RandomState(decim + ntaps)
    taps = {"lpf_div8": lambda: pkg.synth.design_lpf(ntaps, 12500.0, fs) / 8.0,  # every tap fits one byte: no high plane at all
            "lpf": lambda: pkg.synth.design_lpf(ntaps, 12500.0, fs),
            "lpf_x8": lambda: pkg.synth.design_lpf(ntaps, 12500.0, fs) * 8.0,
            "dense": lambda: rng.uniform(-0.3, 0.3, ntaps)}[shape]()
    offs = [25000 * k + (137 if k % 3 == 0 else 0) for k in range(-10, 11)]
    iq = pkg.synth.synth_iq(decim * 400 + ntaps + 7, fs, offs[:3], seed=decim)
    masks = {}
    for kernel, resident in (("auto", 1), ("mfma1", 1), ("mfma1s", 0)):
        eng = _mk_engine(pkg, fs, decim, taps, offs, max_block=1 << 16, want_iq=False, kernel=kernel)
        st = eng.stats()
        eng.close()
        # round 5: "auto" is the second generation's long-filter kernel (mfm_kernel_v3l.hip) where an instance is built for
        # the geometry - all but sixteen k-steps of int16 taps with every high-byte plane held and eight staging chunks per
        # thread on half-tile images (dense taps at decimations of 320 and more), which stay on the first generation
        assert st["taps_resident"] == resident, st
        if kernel != "auto":
            assert st["kernel_variant"] == 1, st
        elif decim <= 200 or (decim <= 400 and shape in ("lpf", "lpf_div8")):
            assert st["kernel_variant"] == 2, (st, decim, ntaps, shape)  # (decimation 448: two half-tile images exceed LDS)
        masks[kernel] = st["tap_hi_mask"]
        _check(pkg, ora, fs, decim, taps, offs, iq, 1 << 16, want_iq=False, kernel=kernel)
        _check(pkg, ora, fs, decim, taps, offs, iq, 30001, want_iq=False, kernel=kernel)
    if shape == "lpf_div8":
        assert masks["auto"] == 0
    if shape == "dense":  # high bytes in every k-step that holds taps: the all-planes instance
        assert masks["auto"] & ~(0x0ff0 if st["k_steps"] == 16 else 0x3c), hex(masks["auto"])
    # a filtered-IQ consumer (signalDebugFile, multifm/demod.c:75-81): since round 6 on the long-filter kernel as well (a run-time
    # switch in its epilogue) wherever that kernel takes the geometry; the first generation's streamed form elsewhere (its
    # resident instances are not built with the IQ store)
    eng = _mk_engine(pkg, fs, decim, taps, offs[:5], max_block=1 << 16, want_iq=False)
    plain = eng.stats()
    eng.close()
    eng = _mk_engine(pkg, fs, decim, taps, offs[:5], max_block=1 << 16, want_iq=True)
    st_iq = eng.stats()
    eng.close()
    assert st_iq["kernel_variant"] == plain["kernel_variant"]
    assert st_iq["taps_resident"] == (1 if plain["kernel_variant"] == 2 else 0), (st_iq, plain)
    _check(pkg, ora, fs, decim, taps, offs[:5], iq, 30001, want_iq=True)


@pytest.mark.parametrize("fmt", [1, 2, 3])
@pytest.mark.parametrize("decim,ntaps,gain", [(96, 512, 1.0), (96, 400, 8.0), (400, 512, 1.0), (200, 500, 30.0), (400, 512, 0.1)])
def test_long_filters_on_8bit_blocks_resident_and_streamed(pkg, ora, fmt, decim, ntaps, gain):
    """The same for 8-bit blocks read as bytes (one sample plane, sixteen k-steps of taps in registers)."""
    fs = 2400000
    taps = pkg.synth.design_lpf(ntaps, 9000.0, fs) * gain
    offs = [25000 * k + (137 if k % 3 == 0 else 0) for k in range(-9, 10)]
    rng = np.random.RandomState(decim + fmt)
    blocks = [(rng.randint(0, 256, size=(m, 2)).astype(np.uint8), fmt) for m in (65536, 4096, 30000, 50000, 12346)]
    got, want, st = _ingest_8bit(pkg, ora, fs, decim, taps, offs, blocks, 65536)
    assert st["kernel_variant"] == 2 and st["launches_8bit"] == st["launches"] > 0 and st["taps_resident"] == 1
    assert got.shape == want.shape and np.array_equal(got, want)
    got1, _, st1 = _ingest_8bit(pkg, ora, fs, decim, taps, offs, blocks, 65536, flags=pkg.binding.MFM_F_FORCE_MFMA_V1)
    assert st1["kernel_variant"] == 1 and st1["launches_8bit"] == st1["launches"] > 0 and st1["taps_resident"] == 1
    assert np.array_equal(got1, want)
    got2, _, st2 = _ingest_8bit(pkg, ora, fs, decim, taps, offs, blocks, 65536, flags=pkg.binding.MFM_F_STREAM_TAPS)
    assert st2["launches_8bit"] == st2["launches"] > 0 and st2["taps_resident"] == 0
    assert np.array_equal(got2, want)


@pytest.mark.parametrize("nch", [130, 192, 320])
def test_work_items_of_a_launch_fit_one_round_of_workgroups(pkg, ora, nch):
    """Round 6: the second-generation kernels deal (chunk, slice) items chunk-major over the eight XCDs; with 3 or 5 slices the
    engine's 'slots / slices' chunks per slice was not a multiple of 8, the grid overflowed the 512 workgroup slots and a few
    workgroups ran a second chunk behind everybody else's only one (a launch took up to twice its work,
    profiles/r06_ab_chunking.txt).  The chunk count is a multiple of 8 now: the last launch's grid is chunks x slices, within the
    slots - and the PCM is still the oracle's."""
    fs, decim, taps, offs, gains = pkg.synth.plan("cfg3_1024ch", nr_channels=nch)
    block = 96 * 64 * 700
    iq = pkg.synth.synth_iq(block + 128, fs, offs[:3], seed=nch)
    eng = _mk_engine(pkg, fs, decim, taps, offs, gains, max_block=block + 128, want_iq=False)
    pcm, _ = eng.run(iq, block + 128)
    st = eng.stats()
    cre, cim, incr = _oracle_tables(eng, nch)
    eng.close()
    nslices = -(-nch // 64)
    assert st["kernel_variant"] == 2 and st["grid_last"] == ((512 // nslices) & ~7) * nslices, st
    ref, _ = ora.run_channels(iq, cre, cim, incr, decim, threads=8)
    assert np.array_equal(pcm, ref)


@pytest.mark.parametrize("nch", [1, 7, 61, 65, 130])
def test_channel_counts_that_do_not_fill_row_blocks(pkg, ora, nch):
    """The matrix kernel works on blocks of 8 channels per wave and 64 per workgroup slice: counts that leave a row
    block, a wave or a whole slice partly empty (clamped row blocks, dump-slot stores) must still be exact."""
    fs, decim, taps, offs, gains = pkg.synth.plan("cfg2_64ch", nr_channels=nch)
    iq = pkg.synth.synth_iq(96 * 700 + 128, fs, offs[:3], seed=nch)
    _check(pkg, ora, fs, decim, taps, offs, iq, 1 << 15, gains=gains, want_iq=(nch == 61))


def test_cfg5_airspy_rate(pkg, ora):
    """BASELINE configs[4] (int16 path): fs 10 MS/s, D=400, 512 taps.  A 62-output tile of 400-sample strides does not
    fit LDS: single-iteration tiles (31 outputs, up to 8 staging chunks per thread) with streamed taps."""
    fs, decim, taps, offs, gains = pkg.synth.plan("cfg5_airspy", nr_channels=24)
    iq = pkg.synth.synth_iq(400 * 900 + 512, fs, offs[:3], seed=24)
    eng = _mk_engine(pkg, fs, decim, taps, offs, gains, max_block=1 << 17, want_iq=False)
    st = eng.stats()
    eng.close()
    assert st["kernel_variant"] == 2 and st["outputs_per_tile"] == 64  # round 5: 64-output tiles from two half-tile images
    _check(pkg, ora, fs, decim, taps, offs, iq, 1 << 17)
    _check(pkg, ora, fs, decim, taps, offs, iq, 1 << 17, want_iq=False)
    eng = _mk_engine(pkg, fs, decim, taps, offs, gains, max_block=1 << 17, want_iq=False, kernel="mfma1")
    st = eng.stats()
    eng.close()
    assert st["kernel_variant"] == 1 and st["outputs_per_tile"] == 31
    _check(pkg, ora, fs, decim, taps, offs, iq, 1 << 17, want_iq=False, kernel="mfma1")


@pytest.mark.parametrize("decim,ntaps", [(256, 256), (200, 256), (320, 512), (136, 160), (176, 192), (400, 400)])
def test_large_decimations_use_single_iteration_tiles(pkg, ora, decim, ntaps):
    fs = 4000000
    taps = pkg.synth.design_lpf(ntaps, 12500.0, fs)
    offs = [12345, -250000, 1000000, -1234567, 31250]
    iq = pkg.synth.synth_iq(decim * 300 + ntaps + 3, fs, offs[:3], seed=decim)
    eng = _mk_engine(pkg, fs, decim, taps, offs, max_block=1 << 16, want_iq=False, kernel="mfma1")
    assert eng.stats()["kernel_variant"] == 1
    eng.close()
    _check(pkg, ora, fs, decim, taps, offs, iq, 1 << 16, want_iq=(decim == 256), kernel="mfma1")
    _check(pkg, ora, fs, decim, taps, offs, iq, 1 << 16, want_iq=False)


@pytest.mark.parametrize("decim,ntaps", [(1, 16), (2, 9), (7, 33), (8, 8), (8, 17), (16, 64), (24, 100), (25, 128),
                                         (40, 128), (96, 96), (97, 128), (104, 128), (128, 128), (200, 256),
                                         (96, 512), (96, 256), (64, 300), (40, 200), (8, 160), (48, 129)])
def test_odd_geometries(pkg, ora, decim, ntaps):
    """decimationFactor 1 (etc/multifm_file.json), taps == decimation, odd tap counts, D not dividing anything."""
    fs = 1000000
    taps = pkg.synth.design_lpf(ntaps, 20000.0, fs)
    offs = [12345, -250000, 100000]
    n = max(decim * 700 + ntaps + 5, 3000)
    iq = pkg.synth.synth_iq(n, fs, offs, seed=decim)
    _check(pkg, ora, fs, decim, taps, offs, iq, 8192)


@pytest.mark.parametrize("decim,ntaps,nch", [(25, 128, 2), (25, 128, 70), (7, 33, 3), (12, 40, 9), (20, 64, 5), (97, 128, 3),
                                             (50, 200, 4), (99, 300, 3), (6, 16, 2), (36, 36, 3), (100, 300, 2)])
def test_decimations_that_are_not_multiples_of_8_run_on_the_matrix_kernel(pkg, ora, decim, ntaps, nch):
    """LDS rows padded to 16-byte multiples, zero taps over the padding, 4-sample staging chunks that straddle rows
    stored sample by sample: any decimation with at least 3/4 of a padded row in use."""
    fs = 1200000
    taps = pkg.synth.design_lpf(ntaps, 12500.0, fs)
    offs = pkg.synth.channel_offsets(nch, fs) if nch > 3 else [12345, -250000, 100000][:nch]
    n = decim * 900 + ntaps + 11
    iq = pkg.synth.synth_iq(n, fs, list(offs)[:3], seed=decim + ntaps)
    eng = _mk_engine(pkg, fs, decim, taps, offs, max_block=1 << 15, want_iq=False)
    st = eng.stats()
    eng.close()
    # decimation 25 with up to 150 taps: the second generation on padded rows (round 4) - unless a channel wants its filtered
    # IQ; every other decimation of this list: the first generation
    # ... and, round 5, filters of 129 taps and more (mfm_kernel_v3l.hip takes any decimation the first generation takes)
    row_bytes = (2 * decim + 15) & ~15
    k_steps = -(-(((ntaps - 1) // decim) * row_bytes + 2 * ((ntaps - 1) % decim) + 2) // 64)  # of 64 elements, padding included
    assert st["kernel_variant"] == (2 if decim == 25 or k_steps >= 5 else 1), st
    _check(pkg, ora, fs, decim, taps, offs, iq, 1 << 15, want_iq=(nch == 2))
    _check(pkg, ora, fs, decim, taps, offs, iq, 5000, want_iq=False)
    if decim == 25:
        _check(pkg, ora, fs, decim, taps, offs, iq, 7777, want_iq=False, kernel="mfma1")


@pytest.mark.parametrize("kernel", KERNELS)
def test_full_scale_random_input_wraps_like_int32(pkg, ora, kernel):
    """Uniform full-scale int16 noise with a wide, high-gain filter overflows the accumulator and hits every
    atan2 octant, including exact ties."""
    fs, decim = 2400000, 96
    taps = pkg.synth.design_lpf(128, 400000.0, fs)
    offs = [0, 101000, -600000, 37500]
    gains = [5.0, 1.0, 4.0, 0.5]
    iq = pkg.synth.random_iq(96 * 3000 + 128, seed=31, full_scale=True)
    iq[5000:6000] = 32767
    iq[7000:8000] = -32768
    iq[9000:9500] = 0  # all-zero stretch: atan2(0, 0) path
    _check(pkg, ora, fs, decim, taps, offs, iq, 1 << 16, gains=gains, kernel=kernel)


def test_taps_too_large_for_byte_split_fall_back_to_dot2(pkg, ora):
    """A tap above 32639 cannot be split into two signed bytes: the engine must pick the v_dot2 kernel and
    still be exact."""
    fs, decim = 2400000, 96
    taps = pkg.synth.design_lpf(128, 400000.0, fs)
    offs = [0, 37500]
    gains = [6.275, 1.0]  # peak tap 5212 * 6.275 = 32705: an int16, but not two signed bytes
    iq = pkg.synth.random_iq(96 * 600 + 128, seed=33, full_scale=True)
    eng = _mk_engine(pkg, fs, decim, taps, offs, gains, max_block=1 << 16)
    assert np.abs(_all_taps(eng, 2)).max() > 32639 and eng.stats()["kernel_variant"] == 0
    cre, cim, incr = _oracle_tables(eng, 2)
    pcm, _ = eng.run(iq, 1 << 16)
    eng.close()
    ref, _ = ora.run_channels(iq, cre, cim, incr, decim)
    assert np.array_equal(pcm, ref)


def test_ragged_blocks_and_blocks_shorter_than_the_filter(pkg, ora):
    """Chunking independence through the engine: 1-sample blocks, blocks < taps (no output), the rtl_sdr /
    file_if / uhd buffer sizes (131072 / 4096 / 16384), a prime size."""
    fs, decim, taps, offs, gains = pkg.synth.plan("cfg2_64ch", nr_channels=9)
    n = 300000
    iq = pkg.synth.synth_iq(n, fs, offs[:3], seed=41)
    eng = _mk_engine(pkg, fs, decim, taps, offs, want_iq=True, max_block=131072)
    cre, cim, incr = _oracle_tables(eng, len(offs))
    ref, refq = ora.run_channels(iq, cre, cim, incr, decim, threads=4, want_iq=True)
    sizes = [1, 1, 100, 26, 1, 4096, 131072, 16384, 7919, 5, 127, 128, 129, 60000, 1000]
    pcm_parts, q_parts, pos, k = [], [], 0, 0
    while pos < n:
        m = min(sizes[k % len(sizes)], n - pos)
        rc = eng.push(iq[pos:pos + m])
        if rc == pkg.binding.MFM_E_BUSY:
            got = eng.fetch()
            pcm_parts.append(got[1])
            q_parts.append(got[2])
            continue
        assert rc == 0, eng.lib.mfm_last_error()
        pos += m
        k += 1
    while True:
        got = eng.fetch()
        if got is None:
            break
        pcm_parts.append(got[1])
        q_parts.append(got[2])
    st = eng.stats()
    eng.close()
    pcm = np.concatenate(pcm_parts, axis=1)
    q = np.concatenate(q_parts, axis=1)
    assert np.array_equal(pcm, ref) and np.array_equal(q, refq)
    assert st["samples_in"] == n and st["outputs"] == ref.shape[1]
    assert st["tail_samples"] == n - ref.shape[1] * decim


@pytest.mark.parametrize("kernel", KERNELS)
def test_long_stream_crosses_rotator_preperiod(pkg, ora, kernel):
    """A stream long enough that the tabulated rotator leaves its pre-period and wraps its cycle several
    times (offset 101 kHz: pre-period 53 105 outputs, period 25; offset 777 Hz: 71 743 / 740)."""
    fs, decim = 2400000, 96
    taps = pkg.synth.design_lpf(128, 12500.0, fs)
    offs = [101000, 777, 3125, 37500]
    n = 96 * 160000 + 128
    iq = pkg.synth.synth_iq(n, fs, offs[:2], seed=51, noise=2000)
    _check(pkg, ora, fs, decim, taps, offs, iq, 1 << 20, want_iq=True, kernel=kernel)


def test_reset_restarts_the_stream(pkg, ora):
    fs, decim, taps, offs, gains = pkg.synth.plan("cfg2_64ch", nr_channels=5)
    iq = pkg.synth.synth_iq(96 * 400 + 200, fs, offs, seed=61)
    eng = _mk_engine(pkg, fs, decim, taps, offs, max_block=1 << 15)
    a, _ = eng.run(iq, 5000)
    eng.reset()
    b, _ = eng.run(iq, 1 << 15)
    eng.close()
    assert np.array_equal(a, b)


def test_device_resident_submit_path(pkg, ora):
    """acquire_input()/submit() with the block already in HBM (what bench.py and an RCCL broadcast use)."""
    import torch
    fs, decim, taps, offs, gains = pkg.synth.plan("cfg2_64ch", nr_channels=16)
    blk = 1 << 16
    iq = pkg.synth.synth_iq(blk * 3, fs, offs[:4], seed=71)
    eng = _mk_engine(pkg, fs, decim, taps, offs, max_block=blk, flags=pkg.binding.MFM_F_DEVICE_ONLY)
    cre, cim, incr = _oracle_tables(eng, len(offs))
    ref, _ = ora.run_channels(iq, cre, cim, incr, decim, threads=4)
    src = torch.from_numpy(iq.reshape(-1)).cuda()
    outs = []
    for b in range(3):
        ptr, cap = eng.acquire_input()
        assert cap >= blk
        hip = torch.cuda.current_stream().cuda_stream
        # device-to-device placement of the block on torch's stream, then submit ordered after it
        import ctypes as C
        rt = C.CDLL("libamdhip64.so")
        rt.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        assert rt.hipMemcpyAsync(ptr, src[b * blk * 2:].data_ptr(), blk * 4, 3, hip) == 0
        eng.submit(blk, producer_stream=hip)
        eng.sync()
        dptr, stride, nout, _ = eng.last_output_device()
        host = np.empty((len(offs), stride), np.int16)
        rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        assert rt.hipMemcpy(host.ctypes.data, dptr, host.nbytes, 2) == 0
        outs.append(host[:, :nout].copy())
    eng.close()
    assert np.array_equal(np.concatenate(outs, axis=1), ref)


@pytest.mark.parametrize("kernel", KERNELS)
def test_full_size_block_properties(pkg, ora, kernel):
    """BASELINE full size (64 channels, 2^24-sample blocks): bit-exact against the threaded oracle, and
    the size-independent property that re-blocking the same stream leaves the PCM unchanged."""
    fs, decim, taps, offs, gains = pkg.synth.plan("cfg2_64ch")
    n = (1 << 24) + 12345
    iq = pkg.synth.synth_iq(n, fs, offs[::9], seed=81)
    big = _check(pkg, ora, fs, decim, taps, offs, iq, 1 << 24, want_iq=False, kernel=kernel)
    eng = _mk_engine(pkg, fs, decim, taps, offs, max_block=1 << 20, kernel=kernel)
    small, _ = eng.run(iq, (1 << 20) - 77)
    eng.close()
    assert np.array_equal(big, small)


def _tiled_iq(pkg, n, fs, carriers, seed):
    """n samples made of a 2^22-sample synthetic base repeated (what bench.py feeds the engine)"""
    base = pkg.synth.synth_iq(1 << 22, fs, carriers, seed=seed)
    return np.tile(base, (-(-n // base.shape[0]), 1))[:n]


@pytest.mark.parametrize("shape", ["cfg2_64ch_2p26", "cfg3_1024ch_2p24", "cfg5_256ch_2p24", "exact_grid_64ch_2p26",
                                   "exact_grid_1024ch_2p24"])
def test_bench_shapes_against_the_oracle(pkg, ora, shape):
    """The sizes the numbers in BENCH_*.json, profiles/ and DESIGN.md are quoted on ('Next round' 2a): one
    block of 2^26 samples for 64 channels (bench.py's default and the driver's line), one block of 2^24 samples for 1024
    channels on one GPU (16 slices x chunked tiles x the XCD item map), 256 channels of the Airspy plan (D = 400, 512 taps)
    - every channel and every output against the oracle, bit-exact, plus the re-blocking property (the same stream in
    smaller, ragged blocks leaves the PCM unchanged).  The exact_grid shapes put every channel on the 25 kHz grid, where
    all rotators are exact and the kernel instances without derotation run (filter/direct_fir.c:151-172)."""
    name, nch, log2 = {"cfg2_64ch_2p26": ("cfg2_64ch", 64, 26), "cfg3_1024ch_2p24": ("cfg3_1024ch", 1024, 24),
                       "cfg5_256ch_2p24": ("cfg5_airspy", 256, 24), "exact_grid_64ch_2p26": ("cfg2_64ch", 64, 26),
                       "exact_grid_1024ch_2p24": ("cfg3_1024ch", 1024, 24)}[shape]
    fs, decim, taps, offs, gains = pkg.synth.plan(name, nr_channels=nch)
    if shape.startswith("exact_grid"):
        # multiples of the output rate (identity rotators) and odd multiples of half of it (sign flips), mixed
        offs = pkg.synth.grid_offsets(nch, fs, decim)
    block = 1 << log2
    n = block + 54321
    iq = _tiled_iq(pkg, n, fs, offs[:: max(1, nch // 6)][:6], seed=log2 + nch)
    threads = os.cpu_count() or 8
    eng = _mk_engine(pkg, fs, decim, taps, offs, gains, max_block=block, want_iq=False)
    st = eng.stats()
    if shape.startswith("exact_grid"):
        assert st["rot_exact_channels"] == nch and st["kernel_variant"] == 2
    cre, cim, incr = _oracle_tables(eng, nch)
    pcm, _ = eng.run(iq, block)
    eng.close()
    ref, _ = ora.run_channels(iq, cre, cim, incr, decim, threads=threads)
    assert pcm.shape == ref.shape
    if not np.array_equal(pcm, ref):
        bad = np.argwhere(pcm != ref)
        raise AssertionError(f"{shape}: {len(bad)} PCM samples differ from the oracle; first at (chan, n) = {bad[0]}")
    del ref
    small_block = (block >> 3) - 4097
    eng = _mk_engine(pkg, fs, decim, taps, offs, gains, max_block=small_block, want_iq=False)
    small, _ = eng.run(iq, small_block)
    eng.close()
    assert np.array_equal(pcm, small), f"{shape}: re-blocking changed the PCM"


@pytest.mark.parametrize("mix", ["all_four", "exact_three", "flip_and_identity", "identity", "quarter_only"])
def test_rotator_classes_mixed_in_one_channel_set(pkg, ora, mix):
    """filter/direct_fir.c:151-172 for increments that make the recurrence exact: (16384, 0) (identity), (-16384, 0) (sign
    flip), (0, +-16384) (quarter turns), next to tabulated ones.  Rows are ordered by class, a launch runs the instance of
    its lowest class and, in a launch with tabulated channels, each wave picks the exact form when its eight channels
    allow it: channel sets that mix the classes at every granularity - within a wave, across waves, across slices - with
    the filtered-IQ output on (which shows the derotated samples themselves) and off, several ragged blocks so that the
    phases cross launches at odd output counts, full-scale input so that components of -32768 reach the sign changes."""
    fs, decim = 2400000, 96
    taps = pkg.synth.design_lpf(128, 12500.0, fs)
    rng = np.random.RandomState(5)
    kinds_of = {"all_four": [0, 1, 2, 3], "exact_three": [0, 1, 2], "flip_and_identity": [0, 1], "identity": [0],
                "quarter_only": [2]}[mix]
    for nch, want_iq in ((3, True), (20, False), (70, True), (200, False)):
        kinds = rng.choice(kinds_of, size=nch)
        k = rng.randint(-40, 40, size=nch)
        offs = np.select([kinds == 0, kinds == 1, kinds == 2], [25000 * k, 12500 * (2 * k + 1), 6250 * (2 * k + 1)],
                         rng.randint(-1100000, 1100000, size=nch)).astype(np.int32)
        iq = pkg.synth.random_iq(96 * 1501 + 333, seed=nch)
        eng = _mk_engine(pkg, fs, decim, taps, offs, max_block=96 * 333 + 17, want_iq=want_iq)
        assert eng.stats()["rot_exact_channels"] == int((kinds != 3).sum())
        eng.close()
        _check(pkg, ora, fs, decim, taps, offs, iq, 96 * 333 + 17, want_iq=want_iq)


def test_oracle_unpack_known_answers(ora):
    """the reference's 8-bit widenings restated (file_if.c:66-157, rtl_sdr_if.c:146-158)."""
    raw = np.array([0x7F, 0x80, 0xFF, 0x01, 0x00, 0x7F, 0x80, 0xFF], np.uint8)
    assert list(ora.unpack_bytes(raw, 1)) == [127, -128, -1, 1, 0, 127, -128, -1]
    assert list(ora.unpack_bytes(raw, 2)) == [0, -255, -128, -126, -127, 0, -255, -128]
    assert list(ora.unpack_bytes(np.array([0, 127, 128, 255], np.uint8), 3)) == [-16256, 0, 128, 16384]
    # three samples = 6 bytes: the last sample falls into the remainder loop and is stored as a bare cast
    assert list(ora.unpack_bytes(raw[:6], 2)) == [0, -255, -128, -126, 0, 127]
    assert list(ora.unpack_bytes(raw[:6], 1)) == [127, -128, -1, 1, 0, 127]


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", [1, 2, 3])
def test_gpu_ingest_of_8bit_formats(pkg, ora, fmt):
    """mfm_engine_push_bytes: raw byte pairs over PCIe, widened on the device; PCM must equal the oracle run on the
    reference's host-side widening of the same reads (odd and short block sizes included)."""
    fs, decim = 1000000, 40
    taps = pkg.synth.design_lpf(64, 12500.0, fs)
    offs = [112500, -200000, 3125]
    rng = np.random.RandomState(fmt)
    sizes = [4096, 4095, 1, 7, 8, 9, 65536, 12345, 4096, 333]
    n = sum(sizes)
    raw = rng.randint(0, 256, size=(n, 2)).astype(np.uint8)
    eng = pkg.Engine(fs, decim, 65536, device=0)
    for o in offs:
        eng.add_channel(int(o), taps, 1.0)
    eng.commit()
    pos, iq, got = 0, [], []
    for m in sizes:
        blk = raw[pos:pos + m]
        iq.append(ora.unpack_bytes(blk, fmt).reshape(-1, 2))
        while True:
            rc = eng.push_bytes(blk, fmt)
            if rc == 0:
                break
            assert rc == pkg.binding.MFM_E_BUSY
            got.append(eng.fetch()[1])
        pos += m
    eng.sync()
    while True:
        b = eng.fetch()
        if b is None:
            break
        got.append(b[1])
    eng.close()
    iq = np.concatenate(iq)
    cre = np.stack([ora.make_taps(taps, int(o), fs, 1.0)[0] for o in offs])
    cim = np.stack([ora.make_taps(taps, int(o), fs, 1.0)[1] for o in offs])
    incr = np.stack([ora.rot_incr(int(o), fs, decim) for o in offs])
    want, _ = ora.run_channels(iq, cre, cim, incr, decim)
    got = np.concatenate(got, axis=1)
    assert got.shape == want.shape and np.array_equal(got, want)


def _ingest_8bit(pkg, ora, fs, decim, taps, offs, blocks, max_block, flags=0):
    """push_bytes a list of (raw uint8 [m][2], MFM_IN_* format) blocks; returns (PCM, oracle PCM, stats)"""
    eng = pkg.Engine(fs, decim, max_block, device=0, flags=flags)
    for o in offs:
        eng.add_channel(int(o), taps, 1.0)
    eng.commit()
    iq, got = [], []
    for blk, fmt in blocks:
        if fmt == 0:
            iq.append(blk.astype(np.int16).reshape(-1, 2))
        else:
            iq.append(ora.unpack_bytes(blk, fmt).reshape(-1, 2))
        while True:
            rc = eng.push(blk.reshape(-1)) if fmt == 0 else eng.push_bytes(blk, fmt)
            if rc == 0:
                break
            assert rc == pkg.binding.MFM_E_BUSY
            got.append(eng.fetch()[1])
    eng.sync()
    while True:
        b = eng.fetch()
        if b is None:
            break
        got.append(b[1])
    st = eng.stats()
    eng.close()
    iq = np.concatenate(iq)
    cre = np.stack([ora.make_taps(taps, int(o), fs, 1.0)[0] for o in offs])
    cim = np.stack([ora.make_taps(taps, int(o), fs, 1.0)[1] for o in offs])
    incr = np.stack([ora.rot_incr(int(o), fs, decim) for o in offs])
    want, _ = ora.run_channels(iq, cre, cim, incr, decim)
    return np.concatenate(got, axis=1), want, st


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", [1, 2, 3])
@pytest.mark.parametrize("geom", ["d96_t128", "d32_t32", "d64_t64", "d128_t128", "d96_t128_gen1", "d25_t128", "d40_t64",
                                  "d96_t512", "d400_t512", "d7_t33", "d25_t170", "d30_t150", "d24_t140"])
def test_gpu_8bit_blocks_read_as_bytes_by_the_matrix_kernel(pkg, ora, fmt, geom):
    """Where the second-generation matrix kernel runs, an 8-bit block stays bytes in HBM and the kernel's GEMM takes
    the one sample plane as it is (mfm_kernel_v3.hip, IN8): the PCM must equal the oracle run on the reference's
    host-side widening (rtl_sdr_if.c:146-158, file_if.c:66-157), block sizes ragged, the history crossing blocks as
    bytes.  Full-range bytes (0x00, 0x7f, 0x80, 0xff all occur)."""
    fs = 2400000
    # second-generation kernel; first-generation kernel: forced, rows that are not multiples of 16 bytes (chunks straddle
    # rows), streamed filters, single-iteration tiles, a tiny decimation
    decim, ntaps = {"d96_t128": (96, 128), "d32_t32": (32, 32), "d64_t64": (64, 64), "d128_t128": (128, 128),
                    "d96_t128_gen1": (96, 128), "d25_t128": (25, 128), "d40_t64": (40, 64), "d96_t512": (96, 512),
                    "d400_t512": (400, 512), "d7_t33": (7, 33),
                    # 7 / 5 / 5 k-steps of taps held in registers (one and two staging chunks per thread)
                    "d25_t170": (25, 170), "d30_t150": (30, 150), "d24_t140": (24, 140)}[geom]
    # d40: chunk-row layout; d25_t128: padded rows (round 4; 170 taps span seven rows and stay on the first generation)
    # round 5: the long filters and the odd decimations whose padded filter spans five k-steps or more run the second
    # generation's long-filter kernel (mfm_kernel_v3l.hip)
    variant = 1 if geom in ("d96_t128_gen1", "d7_t33") else 2
    base_flags = pkg.binding.MFM_F_FORCE_MFMA_V1 if geom == "d96_t128_gen1" else 0
    taps = pkg.synth.design_lpf(ntaps, 9000.0, fs) * (3.0 if geom in ("d96_t512", "d7_t33") else 1.0)
    offs = [25000 * k + (137 if k % 3 == 0 else 0) for k in range(-9, 10)]
    rng = np.random.RandomState(40 + fmt)
    sizes = [65536, 4096, 30000, 2, 8, 96, 50000, 65536, 12346, 332]
    if fmt == 2:
        sizes = [m + (m & 1) for m in sizes]  # cu8 blocks of odd length go the int16 way (next test)
    blocks = []
    for m in sizes:
        raw = rng.randint(0, 256, size=(m, 2)).astype(np.uint8)
        raw[:4] = [[0, 255], [127, 128], [128, 127], [255, 0]][:min(4, m)]
        blocks.append((raw, fmt))
    got, want, st = _ingest_8bit(pkg, ora, fs, decim, taps, offs, blocks, 65536, flags=base_flags)
    assert st["kernel_variant"] == variant
    assert st["launches_8bit"] == st["launches"] > 0
    assert got.shape == want.shape and np.array_equal(got, want)
    # the same through the widening pass: same bits, no byte launches
    got2, _, st2 = _ingest_8bit(pkg, ora, fs, decim, taps, offs, blocks, 65536,
                                flags=base_flags | pkg.binding.MFM_F_WIDEN_8BIT)
    assert st2["launches_8bit"] == 0 and np.array_equal(got2, want)


@pytest.mark.gpu
def test_gpu_8bit_stream_changes_format_mid_way(pkg, ora):
    """A history kept as bytes in front of a block of another format (int16, another 8-bit form, a cu8 block of odd
    length) is widened on the device and the stream goes on as int16; after a block that leaves no history - or an
    engine reset - bytes are read directly again."""
    fs, decim = 2400000, 96
    taps = pkg.synth.design_lpf(128, 9000.0, fs)
    offs = [-300000, 12500, 412500]
    rng = np.random.RandomState(77)

    def raw(m):
        return rng.randint(0, 256, size=(m, 2)).astype(np.uint8)

    def s16(m):
        return rng.randint(-32768, 32768, size=(m, 2)).astype(np.int16)

    blocks = [(raw(20000), 3), (raw(20001), 3), (s16(7777), 0), (raw(9000), 3), (raw(4096), 1), (raw(4097), 2),
              (raw(4098), 2), (raw(10), 1), (s16(96 * 50), 0), (raw(30000), 1)]
    got, want, st = _ingest_8bit(pkg, ora, fs, decim, taps, offs, blocks, 32768)
    assert got.shape == want.shape and np.array_equal(got, want)
    assert 2 <= st["launches_8bit"] < st["launches"]


@pytest.mark.gpu
def test_gpu_device_producer_hands_over_bytes(pkg, ora):
    """mfm_engine_acquire_input_bytes: a producer on the device (here a device-to-device copy standing in for a
    collective) writes 8-bit IQ where the engine says, submit() as for int16 blocks.  Refused with MFM_E_STATE when the
    history in front is of another format, and a cu8 block of odd length is refused at submit."""
    import ctypes as C
    import torch
    rt = C.CDLL("libamdhip64.so")
    rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    fs, decim = 2400000, 96
    taps = pkg.synth.design_lpf(128, 9000.0, fs)
    offs = [-300000, 12500, 412500, 25000]
    rng = np.random.RandomState(3)
    b = pkg.binding
    eng = pkg.Engine(fs, decim, 1 << 15, device=0)
    for o in offs:
        eng.add_channel(int(o), taps, 1.0)
    eng.commit()
    iq, got = [], []
    for m in (30000, 32768, 1000, 20002):
        raw = rng.randint(0, 256, size=(m, 2)).astype(np.uint8)
        iq.append(ora.unpack_bytes(raw, 2).reshape(-1, 2))
        src = torch.from_numpy(raw).to("cuda:0")
        ptr, cap = eng.acquire_input_bytes(b.MFM_IN_CU8)
        assert cap >= m
        assert rt.hipMemcpy(ptr, src.data_ptr(), 2 * m, 3) == 0
        eng.submit(m, producer_stream=0)
        eng.sync()
        got.append(eng.fetch()[1])
    # the history is cu8 bytes now: another format cannot be taken as bytes, and an odd cu8 block not at all
    with pytest.raises(b.MfmError) as ei:
        eng.acquire_input_bytes(b.MFM_IN_RTLSDR_U8)
    assert ei.value.code == b.MFM_E_STATE
    ptr, _ = eng.acquire_input_bytes(b.MFM_IN_CU8)
    with pytest.raises(b.MfmError) as ei:
        eng.submit(777, producer_stream=0)
    assert ei.value.code == b.MFM_E_INVAL
    # ... and the stream is still usable: the same 777 samples widened by the caller, the int16 way
    raw = rng.randint(0, 256, size=(777, 2)).astype(np.uint8)
    wide = ora.unpack_bytes(raw, 2).reshape(-1, 2)
    iq.append(wide)
    assert eng.push(wide.reshape(-1)) == 0
    eng.sync()
    got.append(eng.fetch()[1])
    st = eng.stats()
    eng.close()
    assert st["launches_8bit"] == 4 and st["launches"] == 5
    iq = np.concatenate(iq)
    cre = np.stack([ora.make_taps(taps, int(o), fs, 1.0)[0] for o in offs])
    cim = np.stack([ora.make_taps(taps, int(o), fs, 1.0)[1] for o in offs])
    incr = np.stack([ora.rot_incr(int(o), fs, decim) for o in offs])
    want, _ = ora.run_channels(iq, cre, cim, incr, decim)
    got = np.concatenate(got, axis=1)
    assert got.shape == want.shape and np.array_equal(got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", [1, 3])
def test_gpu_8bit_bytes_on_the_reference_geometry(pkg, ora, fmt):
    """The 2.4 MS/s -> 25 kS/s plan of the headline configuration (64 channels, decimation 96, 128 taps whose outer
    k-steps fit one byte: the kernel instance with compile-time geometry) fed with bytes."""
    fs, decim, taps, offs, gains = pkg.synth.plan("cfg2_64ch")
    rng = np.random.RandomState(50 + fmt)
    eng = pkg.Engine(fs, decim, 1 << 18, device=0)
    for o, g in zip(offs, gains):
        eng.add_channel(int(o), taps, float(g))
    eng.commit()
    iq, got = [], []
    for m in (1 << 18, 100000, 96 * 64 * 7, 262143):
        raw = rng.randint(0, 256, size=(m, 2)).astype(np.uint8)
        iq.append(ora.unpack_bytes(raw, fmt).reshape(-1, 2))
        assert eng.push_bytes(raw, fmt) == 0
        eng.sync()
        got.append(eng.fetch()[1])
    st = eng.stats()
    eng.close()
    assert st["kernel_variant"] == 2 and st["launches_8bit"] == 4
    iq = np.concatenate(iq)
    cre = np.stack([ora.make_taps(taps, int(o), fs, float(g))[0] for o, g in zip(offs, gains)])
    cim = np.stack([ora.make_taps(taps, int(o), fs, float(g))[1] for o, g in zip(offs, gains)])
    incr = np.stack([ora.rot_incr(int(o), fs, decim) for o in offs])
    want, _ = ora.run_channels(iq, cre, cim, incr, decim, threads=8)
    got = np.concatenate(got, axis=1)
    assert got.shape == want.shape and np.array_equal(got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", [2, 3])
def test_full_size_8bit_blocks_bytes_equal_widened(pkg, fmt):
    """BASELINE full size (64 channels, 2^24-sample blocks) fed with 8-bit IQ: the kernel instance that reads the bytes
    (compile-time geometry, hand-scheduled column groups) against the widening pass + int16 kernel on the same stream -
    two independent routes through the engine that must agree sample for sample (each is checked against the oracle at
    small sizes above), and re-blocking the stream must not change the PCM either."""
    fs, decim, taps, offs, gains = pkg.synth.plan("cfg2_64ch")
    rng = np.random.RandomState(90 + fmt)
    n = (1 << 24) + 4322
    raw = rng.randint(0, 256, size=(n, 2)).astype(np.uint8)

    def run(flags, block):
        eng = pkg.Engine(fs, decim, block, device=0, flags=flags)
        for o, g in zip(offs, gains):
            eng.add_channel(int(o), taps, float(g))
        eng.commit()
        out, pos = [], 0
        while pos < n:
            m = min(block, n - pos)
            m -= (m & 1) if fmt == 2 and pos + m < n else 0
            while eng.push_bytes(raw[pos:pos + m], fmt) != 0:
                out.append(eng.fetch()[1])
            pos += m
        eng.sync()
        while True:
            b = eng.fetch()
            if b is None:
                break
            out.append(b[1])
        st = eng.stats()
        eng.close()
        return np.concatenate(out, axis=1), st

    a, sa = run(0, 1 << 24)
    b, sb = run(pkg.binding.MFM_F_WIDEN_8BIT, 1 << 24)
    c, sc = run(0, (1 << 20) - 78)
    assert sa["launches_8bit"] == sa["launches"] and sb["launches_8bit"] == 0 and sc["launches_8bit"] == sc["launches"]
    assert a.shape == b.shape == c.shape and a.shape[1] == (n - len(taps)) // decim + 1
    assert np.array_equal(a, b) and np.array_equal(a, c)


@pytest.mark.gpu
@pytest.mark.parametrize("variant", [0, 1, 2])
def test_discriminator_division_hard_cases_on_the_device(pkg, ora, tmp_path, variant):
    """The three renderings of the discriminator (scalar / packed / four at a time: kernel variants 0 / 1 / 2) on the device,
    through mfm_devtest_discriminate, against the oracle's fm_demod.c:68-72 + fast_atan2f (IEEE division): the quotients
    tools/div_proof.c finds closest to a rounding boundary (a sample of 2^19 of the 46.5 M, every octant), the three that
    depend on the exact v_rcp_f32 result, and 2^22 random int32 pairs."""
    import ctypes as C
    import subprocess
    exe, dump = tmp_path / "div_proof", tmp_path / "hard.bin"
    r = subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-o", str(exe), os.path.join(os.path.dirname(__file__), "..", "tools",
                        "div_proof.c"), "-lm", "-lpthread"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    subprocess.run([str(exe), "1", "8", str(dump), "rn"], capture_output=True, text=True, timeout=600)
    hard = np.fromfile(dump, dtype=np.uint32).reshape(-1, 2).astype(np.int64)
    assert hard.shape[0] == 8 << 16
    crit = np.array([[8388608, 16777215], [13981011, 16777213], [15099490, 16777211]], np.int64)
    rng = np.random.RandomState(variant)
    # scale a share of the hard pairs by powers of two (the operands are int32 converted to float: up to 2^31)
    sh = rng.randint(0, 7, size=hard.shape[0])
    scaled = hard * (1 << sh)[:, None]
    pairs = np.concatenate([crit, crit * 64, hard, scaled])
    re_l, im_l = [], []
    for o in range(8):
        x = pairs[:, 1] if o & 1 else pairs[:, 0]
        y = pairs[:, 0] if o & 1 else pairs[:, 1]
        re_l.append(-x if o & 2 else x)
        im_l.append(-y if o & 4 else y)
    rnd = rng.randint(-2**31, 2**31, size=(2, 1 << 22), dtype=np.int64)
    s_re = np.ascontiguousarray(np.concatenate(re_l + [rnd[0]]), dtype=np.int32)
    s_im = np.ascontiguousarray(np.concatenate(im_l + [rnd[1]]), dtype=np.int32)
    got = np.zeros(s_re.size, np.int16)
    lib = pkg.load_library()
    i32p, i16p = C.POINTER(C.c_int32), C.POINTER(C.c_int16)
    rc = lib.mfm_devtest_discriminate(variant, s_re.ctypes.data_as(i32p), s_im.ctypes.data_as(i32p), s_re.size,
                                      got.ctypes.data_as(i16p), 0)
    assert rc == 0, lib.mfm_last_error()
    want = np.zeros(s_re.size, np.int16)
    ora.lib().mfmo_discriminate_batch(s_re.ctypes.data_as(i32p), s_im.ctypes.data_as(i32p), s_re.size, want.ctypes.data_as(i16p), 0)
    bad = np.flatnonzero(got != want)
    assert bad.size == 0, (bad.size, s_re[bad[:4]], s_im[bad[:4]], got[bad[:4]], want[bad[:4]])


@pytest.mark.gpu
def test_reciprocal_table_is_the_one_the_division_proof_enumerated(pkg):
    """the one-residual-step division is correct for gfx950's v_rcp_f32 (tools/div_proof.c on the table
    tools/rcp_check.hip read off an MI355X).  mfm_devtest_rcp_table hashes what v_rcp_f32 returns for all 2^23 significands on
    THIS device: it must be the table the header names (commit() checks the same and falls back to a 2^28-quotient sweep
    against the IEEE division otherwise), with the proof's shares - 89 % correctly rounded, 9 % one ulp low, 2 % high - and
    the sweep itself must find nothing."""
    import ctypes as C
    lib = pkg.load_library()
    h, bad, tried = C.c_uint64(), C.c_uint64(), C.c_uint64()
    counts = (C.c_uint64 * 4)()
    assert lib.mfm_devtest_rcp_table(0, C.byref(h), counts, C.byref(bad), C.byref(tried)) == 0, lib.mfm_last_error()
    low, exact, high, other = (int(c) for c in counts)
    assert low + exact + high + other == 1 << 23 and other == 0, (low, exact, high, other)
    assert 0.87 < exact / (1 << 23) < 0.91 and 0.07 < low / (1 << 23) < 0.11 and 0.01 < high / (1 << 23) < 0.03, (low, exact, high)
    assert bad.value == 0 and tried.value == 1 << 28, (bad.value, tried.value)
    assert h.value == pkg.binding.MFM_RCP_TABLE_HASH_GFX950, hex(h.value)


@pytest.mark.gpu
@pytest.mark.parametrize("nch", [256, 130, 70])
def test_long_filter_row_block_forms_give_the_same_bits(pkg, ora, nch):
    """mfm_kernel_v3l.hip at configs[4]'s geometry (decimation 400, 512 taps): more than 64 channels run two row blocks per wave
    (128-channel slices, quarter-tile images) unless MFM_F_V3L_ONE_ROW_BLOCK asks for the other form; both, and the first
    generation, against the oracle - int16 blocks of ragged lengths and RTL-SDR bytes."""
    b = pkg.binding
    fs, decim, taps, offs, gains = pkg.synth.plan("cfg5_airspy", nr_channels=nch)
    n = decim * 900 + len(taps) + 17
    iq = pkg.synth.random_iq(n, seed=nch)
    seen = set()
    for flags in (0, b.MFM_F_V3L_ONE_ROW_BLOCK, b.MFM_F_FORCE_MFMA_V1):
        eng = _mk_engine(pkg, fs, decim, taps, offs, gains=gains, max_block=1 << 17, flags=flags)
        st = eng.stats()
        eng.close()
        assert st["kernel_variant"] == (1 if flags == b.MFM_F_FORCE_MFMA_V1 else 2) and st["taps_resident"] == 1, st
        seen.add((st["kernel_variant"], st["lds_bytes"]))
        for block in (1 << 17, 50001):
            eng = _mk_engine(pkg, fs, decim, taps, offs, gains=gains, max_block=block, flags=flags)
            cre, cim, incr = _oracle_tables(eng, len(offs))
            pcm, _ = eng.run(iq, block)
            eng.close()
            ref, _ = ora.run_channels(iq, cre, cim, incr, decim, threads=8)
            assert pcm.shape == ref.shape and np.array_equal(pcm, ref), (flags, block, int((pcm != ref).sum()) if pcm.shape == ref.shape else pcm.shape)
        # the byte form of the same kernels
        raw = np.random.RandomState(nch).randint(0, 256, size=(n, 2)).astype(np.uint8)
        if flags != b.MFM_F_FORCE_MFMA_V1:
            got, want, st8 = _ingest_8bit(pkg, ora, fs, decim, taps, offs, [(raw[:70001], 3), (raw[70001:], 3)], 1 << 19, flags=flags)
            assert st8["kernel_variant"] == 2 and st8["launches_8bit"] > 0 and got.shape == want.shape and np.array_equal(got, want)
    assert len(seen) == 3, seen                       # three different kernels ran (the two long-filter forms differ in their LDS)
