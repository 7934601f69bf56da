"""BCH(31,21) (pager/bch_code.c) and the POCSAG slicer / sync / batch stage
(pager/pager_pocsag.c) - the oracle pinned to what is known about the reference on the CPU, then bit-exact GPU
parity through the C ABI (codewords, BCH verdicts, event positions), alone and behind engine -> resampler."""
import ctypes as C
import itertools
import os

import numpy as np
import pytest

SYNC, IDLE = 0x7CD215D8, 0x6983915E
KAT_CODEWORDS = (0x6983915E, 0x00000000, 0x1BA84B3E)  # idle, zero, bit-reversed sync & 0x7fffffff


# ---- BCH: the oracle against what is known about the reference --------------------------------------------

def test_oracle_bch_field_and_constants(ora, pkg):
    a, idx = ora.bch_tables()
    assert a[:8] == [1, 2, 4, 8, 16, 5, 10, 20] and sorted(a) == list(range(1, 32))  # x^5 = x^2 + 1, primitive
    assert idx[0] == -1 and all(idx[a[i]] == i for i in range(31))
    # the protocol constants of pager_pocsag_priv.h are codewords of the code as the reference maps bits
    for cw in KAT_CODEWORDS:
        assert ora.bch3121_decode(cw) == (0, cw)
    assert pkg.synth.pocsag_codeword(IDLE & 0x1FFFFF) & 0x7FFFFFFF == IDLE
    rev_sync = int(f"{SYNC:032b}"[::-1], 2)
    assert rev_sync & 0x7FFFFFFF == KAT_CODEWORDS[2]
    assert pkg.synth.pocsag_codeword(rev_sync & 0x1FFFFF) == rev_sync  # parity bit included


def test_oracle_bch_error_sweep_matches_reference_counts(ora):
    """SURVEY.md section 8c, recorded from the reference's bch_code_decode: all 31 single and 465 double errors are
    corrected; of the 4495 triples 2480 come back rc 1 with the word untouched, 2015 rc 0 with another codeword."""
    for cw in KAT_CODEWORDS:
        words = [cw ^ (1 << i) for i in range(31)]
        words += [cw ^ (1 << i) ^ (1 << j) for i, j in itertools.combinations(range(31), 2)]
        out, rc = ora.bch3121_decode_batch(words)
        assert not rc.any() and (out == cw).all()
        triples = np.array([cw ^ (1 << i) ^ (1 << j) ^ (1 << k) for i, j, k in itertools.combinations(range(31), 3)],
                           np.uint32)
        out, rc = ora.bch3121_decode_batch(triples)
        untouched = (rc == 1) & (out == triples)
        miscorrected = (rc == 0) & (out != cw)
        assert int(untouched.sum()) == 2480 and int(miscorrected.sum()) == 2015
        assert ((rc == 1) <= (out == triples)).all()  # rc 1 never modifies
        # whatever rc 0 returns is a codeword again
        out2, rc2 = ora.bch3121_decode_batch(out[rc == 0])
        assert not rc2.any() and (out2 == out[rc == 0]).all()


def test_oracle_bch_quirks(ora):
    # bit 31 is never looked at and is carried through (callers mask it, pager_pocsag.c:332)
    assert ora.bch3121_decode(0x80000000 | IDLE) == (0, 0x80000000 | IDLE)
    assert ora.bch3121_decode(0x80000000 | IDLE ^ 0x40) == (0, 0x80000000 | IDLE)
    # S1 == 0 with S3 != 0 falls through with rc 0 and no change (bch_code.c:342,391): find such a word
    a, _ = ora.bch_tables()
    hits = 0
    for e in range(1, 1 << 10):
        w = e << 21  # error pattern in the ten parity positions x^0..x^9 (bits 30..21)
        s1 = 0
        for j in range(10):
            if (w >> (30 - j)) & 1:
                s1 ^= a[j]
        if s1 == 0:
            assert ora.bch3121_decode(w) == (0, w)
            hits += 1
    assert hits == 31  # 2^10 / 2^5 - 1 non-zero patterns with S1 = 0


def test_hosttwin_bch_equals_oracle(ora, pkg):
    """The device's table-driven decode (syndrome bytes -> flip table), run on the host, against the restated
    algorithm: every syndrome class on several codewords (both decoders are functions of the syndrome only), all
    error patterns of weight <= 3 on the KAT codewords, and random words."""
    rng = np.random.RandomState(5)
    words = []
    codewords = list(KAT_CODEWORDS) + [pkg.synth.pocsag_codeword(int(v)) & 0x7FFFFFFF for v in rng.randint(0, 1 << 21, 8)]
    for cw in codewords:
        words += [cw ^ (e << 21) for e in range(1 << 10)]
    for cw in KAT_CODEWORDS:
        for k in (1, 2, 3):
            words += [cw ^ sum(1 << b for b in c) for c in itertools.combinations(range(31), k)]
    words += list(rng.randint(0, 1 << 32, 30000, dtype=np.uint64))
    words = np.array(words, np.uint32)
    want, want_rc = ora.bch3121_decode_batch(words)
    lib = pkg.load_library()
    got = words.copy()
    got_rc = np.zeros(words.size, np.uint8)
    u32p = C.POINTER(C.c_uint32)
    base = got.ctypes.data
    for i in range(words.size):
        got_rc[i] = lib.mfm_hosttwin_bch3121_decode(C.cast(base + 4 * i, u32p))
    assert np.array_equal(got, want) and np.array_equal(got_rc, want_rc)


# ---- POCSAG: oracle self-checks -----------------------------------------------------------------------------

def _messages(sy):
    return [(0x12345, 3, 2, sy.pocsag_alpha_words("HELLO MI355X\x04")),
            (0x00777, 5, 0, sy.pocsag_numeric_words("0123-456 [9]")),
            (0x3FFFF, 0, 3, sy.pocsag_alpha_words("The quick brown fox jumps over the lazy dog 0123456789\x03"))]


def test_oracle_pocsag_decodes_synthetic_pages(ora, pkg):
    sy = pkg.synth
    batches = sy.pocsag_batches(_messages(sy))
    for baud in (512, 1200, 2400):
        pcm = sy.pocsag_pcm(sy.pocsag_bits(batches), baud, noise=1500, lead=5000, trail=9000, seed=baud)
        ev, msgs = ora.Pocsag().feed(pcm)
        types = [int(t) for t in ev["type"]]
        assert types[0] == ora.EV_SYNC_FOUND and types[-1] == ora.EV_SYNC_LOST
        assert types.count(ora.EV_BATCH) == len(batches) and all(int(b) == baud for b in ev["baud"])
        got = [(m[0], m[2], m[3], m[4].rstrip(b"\x00")) for m in msgs]
        # capcode = 18 address bits as they sit in the word << 3 | frame (pager_pocsag.c:362)
        assert got[0] == (2, (0x12345 << 3) | 3, 2, b"HELLO MI355X\x04")
        assert got[1][0] == 3 and got[1][1:3] == ((0x00777 << 3) | 5, 0) and got[1][3].startswith(b"0123-456 [9]")
        assert got[2][:3] == (2, (0x3FFFF << 3) | 0, 3) and got[2][3].startswith(b"The quick brown fox")
        # chunking independence
        o2 = ora.Pocsag()
        ev2, msgs2 = [], []
        for i in range(0, pcm.size, 777):
            e, m = o2.feed(pcm[i:i + 777])
            ev2.append(e)
            msgs2 += m
        assert np.array_equal(np.concatenate(ev2), ev) and msgs2 == msgs


def _uncorrectable_triple(ora, bits, at):
    """three bit positions (0..30) whose inversion in the word starting at bit `at` makes bch_code_decode give up"""
    word = sum(int(bits[at + k]) << k for k in range(31))
    for c in itertools.combinations(range(31), 3):
        w = word ^ sum(1 << b for b in c)
        if ora.bch3121_decode(w) == (1, w):
            return list(c)
    raise AssertionError("no uncorrectable triple")


def test_oracle_message_layer_replays_from_events(ora, pkg):
    """The batch events carry everything the message layer needs: replaying them through the stand-alone message
    decoder gives the same pages as the embedded one (this is how the GPU stage's host side works)."""
    sy = pkg.synth
    bits = sy.pocsag_bits(sy.pocsag_batches(_messages(sy)))
    first = 576 + 32
    flips = [first + 32 * 7 + 3, first + 32 * 7 + 9] + [first + 32 * 8 + b for b in _uncorrectable_triple(ora, bits, first + 32 * 8)]
    pcm = sy.pocsag_pcm(bits, 1200, noise=800, lead=3000, trail=40000, seed=9, flip=flips)
    ev, msgs = ora.Pocsag().feed(pcm)
    dec = ora.PocsagMsgDec()
    replay = []
    for e in ev:
        if e["type"] == ora.EV_BATCH:
            nr_ok, m = dec.batch(e["raw"], int(e["baud"]), int(e["sample"]))
            assert nr_ok == int(e["nr_ok"])
            replay += m
        elif e["type"] == ora.EV_SYNC_LOST:
            replay += dec.flush(int(e["baud"]), int(e["sample"]))
    assert replay == msgs and len(msgs) >= 2
    assert any(int(e["nr_ok"]) < 16 for e in ev if e["type"] == ora.EV_BATCH)  # the triple error ended a batch early


HOST_SO = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tsl-sdr_amd", "host", "libmfm_host.so")


class HostPager:
    """tsl-sdr_amd/host/mfm_pager_pocsag.c through ctypes: collects the pages its callbacks receive."""
    CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint16, C.c_uint32, C.POINTER(C.c_char), C.c_size_t, C.c_uint8)

    def __init__(self):
        if not os.path.exists(HOST_SO):
            pytest.fail(f"{HOST_SO} missing: run make -C tsl-sdr_amd")
        self.h = C.CDLL(HOST_SO)
        self.pages = []
        self._num = self.CB(lambda p, baud, cap, data, n, fn: self._page(3, baud, cap, data, n, fn))
        self._alpha = self.CB(lambda p, baud, cap, data, n, fn: self._page(2, baud, cap, data, n, fn))
        self.p = C.c_void_p()
        self.h.pager_pocsag_new.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, self.CB, self.CB, C.c_bool]
        self.h.pager_pocsag_on_events.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        self.h.pager_pocsag_delete.argtypes = [C.POINTER(C.c_void_p)]
        assert self.h.pager_pocsag_new(C.byref(self.p), 929612500, self._num, self._alpha, False) == 0

    def _page(self, kind, baud, cap, data, n, fn):
        self.pages.append((kind, int(baud), int(cap), int(fn), C.string_at(data, n)))
        return 0

    def on_events(self, ev):
        ev = np.ascontiguousarray(ev)
        assert self.h.pager_pocsag_on_events(self.p, ev.ctypes.data, ev.size) == 0

    def close(self):
        assert self.h.pager_pocsag_delete(C.byref(self.p)) == 0


def _to_product_events(ev, pkg, channel=0):
    out = np.zeros(ev.size, pkg.binding.POCSAG_EVENT_DTYPE)
    for k in ("type", "baud", "aux", "sample", "nr_ok", "fail_mask", "raw", "corrected"):
        out[k] = ev[k]
    out["channel"] = channel
    return out


def test_host_page_assembly_from_events(ora, pkg):
    """The C host's message layer (mfm_pager_pocsag.c), fed the oracle's events in the product's event layout,
    delivers the pages the oracle's embedded message layer delivers - including the early termination on an
    uncorrectable word and the flush when sync is lost."""
    sy = pkg.synth
    assert pkg.binding.POCSAG_EVENT_DTYPE.itemsize == 160 == C.sizeof(pkg.binding.PocsagEvent)
    pcm = _pocsag_channels(sy, ora, 420000)
    for c in range(pcm.shape[0]):
        ev, msgs = ora.Pocsag().feed(pcm[c])
        hp = HostPager()
        half = ev.size // 2
        hp.on_events(_to_product_events(ev[:half], pkg))
        hp.on_events(_to_product_events(ev[half:], pkg))
        hp.close()
        assert hp.pages == [(m[0], m[1], m[2], m[3], m[4]) for m in msgs], f"channel {c}"
    assert len(msgs) > 0


# ---- GPU parity ---------------------------------------------------------------------------------------------

@pytest.mark.gpu
def test_gpu_bch_matches_oracle(ora, pkg):
    rng = np.random.RandomState(11)
    cws = np.array([pkg.synth.pocsag_codeword(int(v)) for v in rng.randint(0, 1 << 21, 2048)], np.uint32)
    classes = (cws[:, None] ^ (np.arange(1 << 10, dtype=np.uint32) << 21)[None, :]).ravel()  # every syndrome class
    sweeps = []
    for cw in KAT_CODEWORDS:
        for k in (1, 2, 3):
            sweeps += [cw ^ sum(1 << b for b in c) for c in itertools.combinations(range(31), k)]
    words = np.concatenate([classes, np.array(sweeps, np.uint32), rng.randint(0, 1 << 32, 1 << 22, dtype=np.uint64).astype(np.uint32),
                            np.zeros(0, np.uint32)])
    want, want_rc = ora.bch3121_decode_batch(words, threads=min(32, os.cpu_count() or 1))
    got, got_rc = pkg.binding.bch3121_decode(words)
    assert np.array_equal(got, want) and np.array_equal(got_rc, want_rc)
    assert int(want_rc.sum()) > 1000 and int((want != words).sum()) > 1000
    # empty input is fine
    e, erc = pkg.binding.bch3121_decode(np.zeros(0, np.uint32))
    assert e.size == 0 and erc.size == 0


def _dedupe(ev, ora):
    """Two detectors can fire on the same sample; the reference lets the later one win (pager_pocsag.c:452-457).
    The oracle logs both SYNC_FOUND events, the GPU stage reports the winner."""
    keep = np.ones(ev.size, bool)
    for i in range(ev.size - 1):
        if ev["type"][i] == ora.EV_SYNC_FOUND and ev["type"][i + 1] == ora.EV_SYNC_FOUND and \
                ev["sample"][i] == ev["sample"][i + 1]:
            keep[i] = False
    return ev[keep]


def _compare_events(got, want, ora, tag):
    want = _dedupe(want, ora)
    assert got.size == want.size, f"{tag}: {got.size} events, oracle {want.size}: " \
        f"{[(int(t), int(s)) for t, s in zip(got['type'], got['sample'])][:12]} vs " \
        f"{[(int(t), int(s)) for t, s in zip(want['type'], want['sample'])][:12]}"
    for k in ("type", "baud", "sample", "aux", "nr_ok", "fail_mask", "raw", "corrected"):
        assert np.array_equal(got[k], want[k]), f"{tag}: field {k} differs at " \
            f"{np.argwhere(np.asarray(got[k] != want[k]).reshape(got.size, -1).any(axis=1)).ravel()[:5]}"


def _pocsag_channels(sy, ora, total):
    """Eight PCM channels of `total` samples covering the cases the state machine has."""
    rng = np.random.RandomState(77)
    msgs = _messages(sy)
    bits = sy.pocsag_bits(sy.pocsag_batches(msgs))
    nb = bits.size

    def place(parts):
        x = np.concatenate(parts)
        assert x.size <= total, (x.size, total)
        pad = rng.normal(0, 900, total - x.size).round().astype(np.int16)
        return np.concatenate([x, pad])

    ch = []
    # 0: 1200 baud, two clean transmissions with a noise gap
    ch.append(place([sy.pocsag_pcm(bits, 1200, noise=700, lead=4000, trail=30000, seed=1),
                     sy.pocsag_pcm(bits, 1200, noise=700, lead=100, trail=5000, seed=2)]))
    # 1: 512 baud with single/double errors in data words and one word with three errors
    first = 576 + 32
    flips = [first + 32 * 1 + 4, first + 32 * 6 + 2, first + 32 * 6 + 29, first + 32 * 7 + 11] + \
        [first + 32 * 9 + b for b in _uncorrectable_triple(ora, bits, first + 32 * 9)] + \
        [first + 544 + 32 * 3 + b for b in (0, 13, 27)]  # second batch: some triple, whatever it decodes to
    ch.append(place([sy.pocsag_pcm(bits, 512, noise=500, lead=7000, trail=50000, seed=3, flip=flips)]))
    # 2: 2400 baud in heavy noise (random bit errors), three times
    ch.append(place([sy.pocsag_pcm(bits, 2400, amplitude=6000, noise=3000, lead=2000 + 700 * k, trail=12000, seed=10 + k)
                     for k in range(3)]))
    # 3: noise only; 4: uniform random samples; 5: silence (all zero -> bit 0 forever)
    ch.append(rng.normal(0, 2000, total).round().astype(np.int16))
    ch.append(rng.randint(-32768, 32768, total).astype(np.int16))
    ch.append(np.zeros(total, np.int16))
    # 6: transmission that starts at sample 0 (no lead) and another one cut off by the end of the capture
    a = sy.pocsag_pcm(bits, 1200, noise=300, lead=0, trail=20000, seed=20)
    b = sy.pocsag_pcm(bits, 512, noise=300, lead=0, trail=0, seed=21)
    x = np.concatenate([a, b])[:total]
    ch.append(np.concatenate([x, np.zeros(total - x.size, np.int16)]) if x.size < total else x)
    # 7: sync word of the second batch with 4 bit errors (kept), a later one with 5 (lost mid-transmission)
    long_bits = sy.pocsag_bits(sy.pocsag_batches(msgs + msgs + msgs))
    s2 = 576 + 544  # first bit of the second sync word
    s3 = 576 + 2 * 544
    flips = [s2 + 1, s2 + 8, s2 + 20, s2 + 31, s3 + 0, s3 + 3, s3 + 9, s3 + 17, s3 + 25]
    ch.append(place([sy.pocsag_pcm(long_bits, 2400, noise=200, lead=3000, trail=30000, seed=30, flip=flips)]))
    assert nb > 0
    return np.stack(ch)


@pytest.mark.gpu
def test_gpu_pocsag_events_match_oracle(ora, pkg):
    sy = pkg.synth
    total = 420000
    pcm = _pocsag_channels(sy, ora, total)
    nch = pcm.shape[0]
    want = []
    for c in range(nch):
        ev, _ = ora.Pocsag().feed(pcm[c])
        want.append(ev)
    assert sum(int((w["type"] == ora.EV_BATCH).sum()) for w in want) >= 20
    for max_in, sizes in ((20000, [1, 7, 1024, 4096, 100, 20000, 333, 20000, 20000, 2048, 31]), (total, [total])):
        gpu = pkg.Pocsag(nch, max_in, device=0)
        got = [[] for _ in range(nch)]
        pos, k = 0, 0
        while pos < total:
            m = min(sizes[k % len(sizes)], total - pos)
            ev = gpu.process_host(pcm[:, pos:pos + m])
            for c in range(nch):
                got[c].append(ev[ev["channel"] == c])
            pos += m
            k += 1
        gpu.close()
        for c in range(nch):
            _compare_events(np.concatenate(got[c]), want[c], ora, f"max_in={max_in} channel {c}")


@pytest.mark.gpu
def test_gpu_chain_iq_to_pocsag_codewords(ora, pkg):
    """BASELINE configs[3]: etc/pocsag_rtlsdr.json front end (fs 1.2 MS/s, D 25 -> 48 kS/s, channels at -320 kHz
    with dBGain 4.0 and -492 kHz), 2-FSK +-4.5 kHz POCSAG bursts at 1200 and 512 bd, PCM -> 4/5 resampler ->
    38 400 Hz -> slicer / sync / BCH, everything after the IQ upload staying in HBM.  Pass = identical corrected
    codewords, BCH verdicts and event positions against the oracle chain, and the pages decode."""
    sy = pkg.synth
    fs, decim, taps, offs, gains = sy.plan("pocsag_rtlsdr")
    msgs = _messages(sy)
    bits = sy.pocsag_bits(sy.pocsag_batches(msgs[:2]))
    n = 3 << 20  # 2.6 s of air time: the 512 bd burst needs 60000 + 1120 * 2344 samples
    lead = 60000
    iq = np.zeros((n, 2), np.int64)
    for o, baud, seed in zip(offs, (1200, 512), (1, 2)):
        b = bits if baud == 1200 else bits[:576 + 544]
        burst = sy.pocsag_fm_iq(b, baud, fs, float(o), lead=lead, trail=0, amplitude=7000.0, noise=150.0, seed=seed)
        m = min(n, burst.shape[0])
        iq[:m] += burst[:m]
        if m < n:  # unmodulated carrier afterwards
            t = np.arange(m, n)
            ph = 2 * np.pi * float(o) * t / fs
            iq[m:, 0] += (7000.0 * np.cos(ph)).round().astype(np.int64)
            iq[m:, 1] += (7000.0 * np.sin(ph)).round().astype(np.int64)
    iq = np.clip(iq, -32768, 32767).astype(np.int16)
    blk = 1 << 19
    eng = pkg.Engine(fs, decim, blk, device=0, flags=pkg.binding.MFM_F_DEVICE_ONLY)
    for o, g in zip(offs, gains):
        eng.add_channel(int(o), taps, float(g))
    eng.commit()
    rtaps = ora.quantize_taps(sy.design_lpf(81, 0.45 / 5, 1.0) * 4)
    rs = pkg.Resampler(len(offs), rtaps, 4, 5, blk // decim + 8, device=0)
    pg = pkg.Pocsag(len(offs), rs.max_out(), device=0)
    got = [[] for _ in offs]
    for b in range(n // blk):
        assert eng.push(iq[b * blk:(b + 1) * blk]) == 0
        dptr, stride, nout, _ = eng.last_output_device()
        yptr, ystride, ny = rs.process_device(dptr, stride, nout, stream=eng.stream)
        pg.process_device(yptr, ystride, ny, stream=eng.stream)
        ev = pg.fetch_events()
        for c in range(len(offs)):
            got[c].append(ev[ev["channel"] == c])
    eng.close()
    rs.close()
    pg.close()
    cre = np.stack([ora.make_taps(taps, int(o), fs, float(g))[0] for o, g in zip(offs, gains)])
    cim = np.stack([ora.make_taps(taps, int(o), fs, float(g))[1] for o, g in zip(offs, gains)])
    incr = np.stack([ora.rot_incr(int(o), fs, decim) for o in offs])
    pcm, _ = ora.run_channels(iq, cre, cim, incr, decim)
    for c in range(len(offs)):
        p384 = ora.Resampler(rtaps, 4, 5).feed(pcm[c])
        want, pages = ora.Pocsag().feed(p384)
        _compare_events(np.concatenate(got[c]), want, ora, f"chain channel {c}")
        assert int((want["type"] == ora.EV_BATCH).sum()) >= 1
        if c == 0:
            assert [m[4].rstrip(b"\x00") for m in pages][:1] == [b"HELLO MI355X\x04"]


def _json_lines(pages):
    """decoder/decoder.c:264-318 with the clock pinned to the epoch (MFM_DECODER_FIXED_TIME)"""
    esc = {0x0A: "\\n", 0x0D: "\\n", 0x22: '\\"', 0x5C: "\\\\", 0x2F: "\\/", 0x08: "<BKSP>", 0x0C: "<FF>", 0x09: "\\t",
           0x03: " ", 0x04: " ", 0x17: " "}
    out = []
    for kind, baud, cap, fn, text, _ in pages:
        body = "".join(esc.get(ch, chr(ch) if 0x20 <= ch < 0x7F else "\\u%04x" % ch) for ch in text)
        out.append('{"proto":"pocsag","type":"%s","timestamp":"1970-01-01 00:00:00 UTC","baud":%d,"capCode":%d,'
                   '"function":%d,"message":"%s"}\n' % ("alphanumeric" if kind == 2 else "numeric", baud, cap, fn, body))
    return "".join(out)


@pytest.mark.gpu
@pytest.mark.parametrize("opts", [[], ["-i", "-b", "-p", "0.999"]])
def test_decoder_amd_json_matches_oracle(tmp_path, ora, pkg, opts):
    """the decoder-shaped driver (decoder/decoder.c:580-673 loop, :264-318 output) on
    three 48 kS/s PCM files -> 4/5 -> 38 400 Hz -> POCSAG; the JSON lines must be what the oracle chain's pages
    print as."""
    import json
    import subprocess
    sy = pkg.synth
    tool = os.path.join(os.path.dirname(HOST_SO), "decoder_amd")
    msgs = _messages(sy) + [(0x2AAAA, 7, 1, sy.pocsag_alpha_words('quote " slash / back \\ tab\t nl\n bell\x07 end\x17'))]
    bits = sy.pocsag_bits(sy.pocsag_batches(msgs))
    total = 900000
    chans = []
    for baud, seed in ((1200, 1), (2400, 2), (512, 3)):
        x = sy.pocsag_pcm(bits, baud, noise=900, lead=9000 + 111 * seed, trail=40000, seed=seed, rate=48000)
        x = np.concatenate([x, np.random.RandomState(seed).normal(0, 900, total).round().astype(np.int16)])[:total]
        chans.append(x)
    taps = sy.design_lpf(81, 0.45 / 5, 1.0) * 4
    (tmp_path / "filter.json").write_text(json.dumps({"lpfCoeffs": [float(t) for t in taps]}))
    invert = "-i" in opts
    paths = []
    for c, x in enumerate(chans):
        p = tmp_path / f"ch{c}.pcm"
        p.write_bytes(((-x.astype(np.int32)).astype(np.int16) if invert else x).tobytes())
        paths.append(str(p))
    out = tmp_path / "pages.json"
    env = dict(os.environ, MFM_DECODER_FIXED_TIME="1")
    r = subprocess.run([tool, "-I", "4", "-D", "5", "-S", "48000", "-F", str(tmp_path / "filter.json"), "-f", "929612500",
                        "-m", "POCSAG", "-c", "-o", str(out), "-B", "100000"] + opts + paths,
                       capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    rtaps = ora.quantize_taps(taps)
    for c, x in enumerate(chans):
        xin = (-x.astype(np.int32)).astype(np.int16) if invert else x
        res = ora.Resampler(rtaps, 4, 5, dc_pole=0.999 if "-b" in opts else None, invert=invert)
        _, pages = ora.Pocsag().feed(res.feed(xin))
        assert len(pages) >= 3
        assert (tmp_path / f"pages.json.{c}").read_text() == _json_lines(pages), f"channel {c}"


@pytest.mark.gpu
def test_pocsag_rtlsdr_chain_through_both_binaries(tmp_path, ora, pkg):
    """BASELINE configs[3] end to end at the reference's process boundary: a cu8 capture (what an RTL-SDR records)
    -> multifm_amd with etc/pocsag_rtlsdr.json's values (fs 1.2 MS/s, D 25, channels at -320 kHz with dBGain 4.0 and
    -492 kHz with the file's "dbGain" typo, i.e. gain 1) -> one PCM sink per channel (the FIFOs) -> decoder_amd
    (4/5 -> 38 400 Hz -> POCSAG) -> JSON lines.  Must equal the oracle chain run on the same bytes: file_if's cu8
    widening, channel path, resampler, pager, decoder.c's output format."""
    import json
    import subprocess
    sy = pkg.synth
    fs, decim, taps, offs, gains = sy.plan("pocsag_rtlsdr")
    center = 929612500 + 320000
    msgs = _messages(sy)
    bits = sy.pocsag_bits(sy.pocsag_batches(msgs[:2]))
    n = 4096 * 700 + 1001  # 2.4 s; a partial, odd-sized last read
    acc = np.zeros((n, 2), np.float64)
    for o, baud, seed in zip(offs, (1200, 2400), (1, 2)):
        burst = sy.pocsag_fm_iq(bits, baud, fs, float(o), lead=50000, trail=0, amplitude=40.0, noise=1.5, seed=seed)
        m = min(n, burst.shape[0])
        acc[:m] += burst[:m]
        if m < n:
            t = np.arange(m, n)
            ph = 2 * np.pi * float(o) * t / fs
            acc[m:, 0] += 40.0 * np.cos(ph)
            acc[m:, 1] += 40.0 * np.sin(ph)
    raw = np.clip(np.round(acc + 127.0), 0, 255).astype(np.uint8)  # 8-bit offset binary, as the dongle delivers
    cap = tmp_path / "capture.cu8"
    cap.write_bytes(raw.tobytes())
    taps_file = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "etc", "lpf_25khz_1200k_128.json")
    lpf = np.array(json.load(open(taps_file))["lpfTaps"])
    cfg = {"device": {"type": "file", "filename": str(cap), "fileFormat": "cu8"}, "sampleRateHz": fs,
           "centerFreqHz": center, "nrSampBufs": 32, "decimationFactor": decim,
           "channels": [{"outFifo": str(tmp_path / "ch0.pcm"), "chanCenterFreq": int(center + offs[0]), "dBGain": 4.0},
                        {"outFifo": str(tmp_path / "ch1.pcm"), "chanCenterFreq": int(center + offs[1]), "dbGain": 4.0}]}
    for c in range(2):
        (tmp_path / f"ch{c}.pcm").write_bytes(b"")
    (tmp_path / "cfg.json").write_text(json.dumps(cfg))
    host_dir = os.path.dirname(HOST_SO)
    r = subprocess.run([os.path.join(host_dir, "multifm_amd"), str(tmp_path / "cfg.json"), taps_file],
                       capture_output=True, text=True, timeout=180)
    assert r.returncode == 0, r.stderr[-2000:]
    rt = sy.design_lpf(81, 0.45 / 5, 1.0) * 4
    (tmp_path / "filter.json").write_text(json.dumps({"lpfCoeffs": [float(t) for t in rt]}))
    env = dict(os.environ, MFM_DECODER_FIXED_TIME="1")
    r = subprocess.run([os.path.join(host_dir, "decoder_amd"), "-I", "4", "-D", "5", "-S", "48000", "-F",
                        str(tmp_path / "filter.json"), "-f", str(center), "-m", "POCSAG", "-c", "-o",
                        str(tmp_path / "pages.json"), str(tmp_path / "ch0.pcm"), str(tmp_path / "ch1.pcm")],
                       capture_output=True, text=True, timeout=180, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    # the oracle chain on the same bytes
    iq = np.concatenate([ora.unpack_bytes(raw[i:i + 4096], 2) for i in range(0, n, 4096)]).reshape(-1, 2)
    g = [10.0 ** (4.0 / 10.0), 1.0]  # multifm/receiver.c:218-220 reads "dBGain" only
    cre = np.stack([ora.make_taps(lpf, int(o), fs, gg)[0] for o, gg in zip(offs, g)])
    cim = np.stack([ora.make_taps(lpf, int(o), fs, gg)[1] for o, gg in zip(offs, g)])
    incr = np.stack([ora.rot_incr(int(o), fs, decim) for o in offs])
    pcm, _ = ora.run_channels(iq, cre, cim, incr, decim)
    rtaps = ora.quantize_taps(rt)
    for c in range(2):
        got_pcm = np.frombuffer((tmp_path / f"ch{c}.pcm").read_bytes(), dtype=np.int16)
        assert np.array_equal(got_pcm, pcm[c]), f"channel {c}: PCM sink differs"
        _, pages = ora.Pocsag().feed(ora.Resampler(rtaps, 4, 5).feed(pcm[c]))
        assert len(pages) >= 2, f"channel {c}: the synthetic pages did not decode"
        assert (tmp_path / f"pages.json.{c}").read_text() == _json_lines(pages), f"channel {c}: JSON lines differ"
