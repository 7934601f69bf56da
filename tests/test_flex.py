"""the FLEX decoder (pager/pager_flex.c).  The oracle restatement is checked against the
reference's protocol constants and against frames built by an independent synthesiser (tsl-sdr_amd/synth.py), the C
host's message walk against the oracle's on clean, damaged and random phases, and the GPU stage (sync 1, FIW, sync 2,
slicing, de-interleave) bit-exact against the oracle through the C ABI - alone, chunked, and end to end into pages."""
import ctypes as C
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST_SO = os.path.join(ROOT, "tsl-sdr_amd", "host", "libmfm_host.so")

RECORDS = [
    dict(kind="alnum", capcode=123456, text="HELLO FLEX WORLD"),
    dict(kind="numeric", capcode=77, digits="5551234"),
    dict(kind="tone", capcode=99, digits="123"),
    dict(kind="siv", capcode=555, siv_type=0, data=0x155),
    dict(kind="alnum", long=(0x1234, 0x1FFFF0), text="long address page", maildrop=True),
    dict(kind="numeric", long=(0x22, 0x1FFF00), digits="0123456789-[]U X 42"),
    dict(kind="tone", long=(0x77, 0x1FFF01), digits="98765432"),
    dict(kind="alnum", capcode=0x1E0000 - 32768, text="fragment", seq=1, fragment=True),   # the last short address
    dict(kind="raw", capcode=4242, vtype=6, body=[0x12345, 0x0ABCDE]),
    dict(kind="tone", capcode=31337, digits="000", ttype=1),
]


def _expected(records, baud, phase, cycle, frame):
    """what a decoder must deliver for flex_phase_words(records): written from the inputs, not from any decoder"""
    out = []
    for r in records:
        if "long" in r:
            cap = (0x1F9001 + (((0x1FFFFF - r["long"][1]) * 32768) + r["long"][0] - 1)) & 0xFFFFFFFF
        else:
            cap = r["capcode"]
        k = r["kind"]
        if k == "alnum":
            aux = int(r.get("fragment", False)) | int(r.get("maildrop", False) and r.get("seq", 3) == 3) << 1 | r.get("seq", 3) << 2
            out.append((1, baud, phase, cycle, frame, aux, 0, 0, cap, r["text"].encode()))
        elif k == "numeric":
            nbits = 2 + 4 * len(r["digits"])
            nwords = (nbits + 20) // 21
            ndig = (21 * nwords - 2) // 4
            out.append((2, baud, phase, cycle, frame, 0, 0, 0, cap, (r["digits"] + " " * ndig)[:ndig].encode()))
        elif k == "tone" and r.get("ttype", 0) == 0:
            d = r["digits"]
            text = d[:3] + ((d[3:] + " " * 5)[:5] if "long" in r else "")
            out.append((2, baud, phase, cycle, frame, 0, 0, 0, cap, text.encode()))
        elif k == "siv":
            out.append((3, baud, phase, cycle, frame, r["siv_type"], r["data"], 0, cap, b""))
    return out


def _pages(msgs):
    """callbacks only (alnum / numeric / siv), without the sample index"""
    return [m[:10] for m in msgs if m[0] <= 3]


# ---- the oracle against the constants and the synthesiser ----------------------------------------------------

def test_oracle_flex_constants_and_word_layout(ora, pkg):
    sy = pkg.synth
    # the A codes of pager_flex.c:46-96 and their frame geometry: every coding fills the same 1.76 s with its blocks,
    # sync 2 lasts 25 ms, and the symbols carry 88 words of 32 bits per phase
    for i, c in enumerate(sy.FLEX_CODINGS):
        o = ora.flex_coding(i)
        assert (o.seq_a, o.baud, o.fsk_levels, o.nr_phases) == (c["seq_a"], c["baud"], c["levels"], len(c["phases"]))
        assert o.symbols_per_block * 100 == 176 * c["sym_rate"] and 16000 // c["sym_rate"] == o.sample_skip + 1
        assert o.symbols_per_block * o.sym_bits == 88 * 32 * o.nr_phases
        assert 2 * (o.sync_2_samples + 16 // o.sym_bits) * (o.sample_skip + 1) == 400
    assert ora.flex_coding(4) is None
    # the mode codes are at Hamming distance >= 8 from one another, so "fewer than 4 differing bits" is unambiguous
    codes = [c["seq_a"] for c in sy.FLEX_CODINGS]
    assert min(bin(a ^ b).count("1") for i, a in enumerate(codes) for b in codes[i + 1:]) >= 8
    # information words carry a nibble checksum of 15; BCH words are codewords of the (31,21) code
    for w in (sy.flex_fiw(3, 17), sy.flex_fiw(15, 127, 1, 1, 15), sy.flex_biw(9, eob=2, priority=3)):
        assert ora.bch3121_decode(w & 0x7FFFFFFF) == (0, w & 0x7FFFFFFF)
        assert sum((w >> (4 * n)) & 0xF for n in range(5)) + ((w >> 20) & 1) & 0xF == 0xF
        assert bin(w).count("1") % 2 == 0


@pytest.mark.parametrize("coding", [0, 1, 2, 3])
def test_oracle_decodes_independent_frames(ora, pkg, coding):
    sy = pkg.synth
    c = sy.FLEX_CODINGS[coding]
    phases, want = {}, []
    for p in c["phases"]:
        recs = [dict(r) for r in RECORDS]
        recs[0]["text"] += " %c" % (65 + p)
        phases[p] = sy.flex_phase_words(recs)
        want += _expected(recs, c["baud"], p, 5, 40 + coding)
    x = sy.flex_pcm([sy.flex_frame_levels(coding, 5, 40 + coding, phases)], lead=777, trail=400, noise=250, seed=coding)
    ev, msgs = ora.Flex().feed(x)
    assert [int(e["type"]) for e in ev] == [ora.FLEX_EV_FRAME]
    e = ev[0]
    assert (int(e["coding"]), int(e["cycle"]), int(e["frame"]), int(e["eye"])) == (coding, 5, 40 + coding, 10)
    assert int(e["sample"]) == 777 + 29995 + (2 if c["sym_rate"] == 3200 else 0)
    assert int(e["a"]) == (c["seq_a"] << 16 | 0x5939) and int(e["b"]) == 0x5555 and int(e["inv_a"]) == int(e["a"]) ^ 0xFFFFFFFF
    assert abs(int(e["sample_range"]) - 18000) < 200 and abs(int(e["sample_delta"])) < 250
    for p in c["phases"]:
        assert np.array_equal(e["words"][p], phases[p])
    assert _pages(msgs) == want
    notes = [m for m in msgs if m[0] > 3]
    assert sorted({m[0] for m in notes}) == [23, 24]            # the hex vector and the sourced tone are only logged


def test_oracle_streaming_and_resets(ora, pkg):
    """chunk boundaries do not matter; damaged sync words end in the events the reference logs"""
    sy = pkg.synth
    ph = {0: sy.flex_phase_words(RECORDS[:3])}
    good = sy.flex_frame_levels(0, 1, 2, ph)
    bad_a = sy.flex_frame_levels(1, 1, 3, {}, a_flip=0x0F0F0000)         # 8 bits of the mode code wrong
    bad_fiw = sy.flex_frame_levels(2, 1, 4, {}, fiw_flip=0x00700000)     # three parity bits: uncorrectable
    x = sy.flex_pcm([good, bad_a, bad_fiw, good], lead=100, trail=3000, noise=500, seed=4, gap=50)
    ev, msgs = ora.Flex().feed(x)
    assert [int(e["type"]) for e in ev] == [1, 2, 3, 1]
    assert int(ev[2]["fiw_rc"]) == 1 and int(ev[1]["coding"]) == 0xFFFFFFFF
    # an idle 1600 bit/s phase filled with alternating all-zero / all-one words is 1010.. on the air: the search
    # locks onto it again and again and reports an unknown mode code every 112 bits (what the reference would log)
    alt = sy.flex_frame_levels(0, 1, 2, {0: sy.flex_phase_words([], idle=(0, 0x1FFFFF))})
    ev_alt, _ = ora.Flex().feed(sy.flex_pcm([alt, good], lead=100, trail=3000, noise=500, seed=5))
    assert [int(e["type"]) for e in ev_alt][:2] == [1, 1] or sum(int(e["type"]) == 2 for e in ev_alt) > 10
    f = ora.Flex()
    ev2, msgs2, pos = [], [], 0
    for n in [1, 309, 311, 5000, 1117, 30000, 7, 64000, 10 ** 9]:
        e, m = f.feed(x[pos:pos + n])
        ev2 += list(e)
        msgs2 += m
        pos += n
    assert len(ev2) == len(ev) and all(a.tobytes() == b.tobytes() for a, b in zip(ev, ev2)) and msgs2 == msgs
    # a checksum error needs a codeword: flip a cycle bit and re-encode the parity
    fiw = sy.flex_codeword((sy.flex_fiw(1, 5) & 0x1FFFFF) ^ 0x10)
    runs = sy.flex_frame_levels(0, 1, 5, {}, fiw_flip=sy.flex_fiw(1, 5) ^ fiw)
    ev3, _ = ora.Flex().feed(sy.flex_pcm([runs], lead=50, trail=2000))
    assert [(int(e["type"]), int(e["fiw_rc"])) for e in ev3] == [(3, 2)]


def test_oracle_corrects_two_errors_per_word(ora, pkg):
    sy = pkg.synth
    recs = RECORDS[:6]
    clean = sy.flex_phase_words(recs)
    rng = np.random.RandomState(3)
    corrupt = {}
    for w in range(88):
        bits = rng.choice(31, 2, replace=False)
        corrupt[(0, w)] = int(1 << bits[0] | 1 << bits[1])
    x = sy.flex_pcm([sy.flex_frame_levels(0, 9, 9, {0: clean}, corrupt=corrupt)], lead=400, trail=400)
    ev, msgs = ora.Flex().feed(x)
    assert _pages(msgs) == _expected(recs, 1600, 0, 9, 9)
    assert not np.array_equal(ev[0]["words"][0], clean)
    # three errors in the block information word: the phase is skipped (pager_flex.c:1122-1127)
    x = sy.flex_pcm([sy.flex_frame_levels(0, 9, 9, {0: clean}, corrupt={(0, 0): 0x7})], lead=400, trail=400)
    ev, msgs = ora.Flex().feed(x)
    assert [m[0] for m in msgs] in ([16], [17], [18])


# ---- the C host's message walk against the oracle's ----------------------------------------------------------

class HostFlex:
    """tsl-sdr_amd/host/mfm_pager_flex.c through ctypes: collects what its callbacks and its note hook receive,
    in the tuple layout of oracle_lib.flex_msg_tuple (without the sample)."""
    ALN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint16, C.c_uint8, C.c_uint8, C.c_uint8, C.c_uint64, C.c_bool, C.c_bool,
                      C.c_uint8, C.POINTER(C.c_char), C.c_size_t)
    NUM = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint16, C.c_uint8, C.c_uint8, C.c_uint8, C.c_uint64, C.POINTER(C.c_char),
                      C.c_size_t)
    SIV = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint16, C.c_uint8, C.c_uint8, C.c_uint8, C.c_uint64, C.c_uint8, C.c_uint32)
    NOTE = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_uint8, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32)

    def __init__(self):
        if not os.path.exists(HOST_SO):
            pytest.fail(f"{HOST_SO} missing: run make -C tsl-sdr_amd")
        self.h = C.CDLL(HOST_SO)
        self.out = []
        self.ctx = (0, 0, 0)
        self._aln = self.ALN(lambda f, baud, ph, cy, fr, cap, frag, md, seq, data, n: self._put(
            (1, baud, ph, cy, fr, int(frag) | int(md) << 1 | seq << 2, 0, 0, cap, C.string_at(data, n))))
        self._num = self.NUM(lambda f, baud, ph, cy, fr, cap, data, n: self._put(
            (2, baud, ph, cy, fr, 0, 0, 0, cap, C.string_at(data, n))))
        self._siv = self.SIV(lambda f, baud, ph, cy, fr, cap, t, d: self._put((3, baud, ph, cy, fr, t, d, 0, cap, b"")))
        self._note = self.NOTE(lambda f, kind, ph, cap, a0, a1, a2: self._put(
            (kind, self.ctx[0], ph, self.ctx[1], self.ctx[2], a0, a1, a2, cap, b"")))
        self.p = C.c_void_p()
        self.h.pager_flex_new.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, self.ALN, self.NUM, self.SIV]
        self.h.pager_flex_on_events.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        self.h.pager_flex_process_phase.argtypes = [C.c_void_p, C.c_void_p, C.c_uint16, C.c_uint8, C.c_uint8, C.c_uint8]
        self.h.pager_flex_set_note_hook.argtypes = [C.c_void_p, self.NOTE]
        self.h.pager_flex_set_note_hook.restype = None
        self.h.pager_flex_delete.argtypes = [C.POINTER(C.c_void_p)]
        assert self.h.pager_flex_new(C.byref(self.p), 929612500, self._aln, self._num, self._siv) == 0
        self.h.pager_flex_set_note_hook(self.p, self._note)

    def _put(self, t):
        self.out.append(tuple(int(v) if not isinstance(v, bytes) else v for v in t))
        return 0

    def phase(self, words, baud, phase, cycle, frame):
        w = np.ascontiguousarray(words, dtype=np.uint32).copy()
        self.ctx = (baud, cycle, frame)
        assert self.h.pager_flex_process_phase(self.p, w.ctypes.data, baud, phase, cycle, frame) == 0
        return w

    def on_events(self, ev, frames):
        ev, frames = np.ascontiguousarray(ev), np.ascontiguousarray(frames)
        for e in ev:                                   # the note hook does not carry the frame context: one by one
            self.ctx = (int(e["baud"]), int(e["cycle"]), int(e["frame"]))
            one = np.array([e])
            assert self.h.pager_flex_on_events(self.p, one.ctypes.data, 1, frames.ctypes.data) == 0

    def close(self):
        assert self.h.pager_flex_delete(C.byref(self.p)) == 0


def _random_phase(sy, rng):
    """a phase whose fields are plausible but arbitrary: exercises every branch of the walk, the error ones included"""
    kind = rng.randint(4)
    if kind == 0:        # valid records, then damage
        n = rng.randint(1, len(RECORDS) + 1)
        recs = [RECORDS[i] for i in rng.choice(len(RECORDS), n, replace=False)]
        w = sy.flex_phase_words(recs, eob=rng.randint(0, 3) if rng.rand() < 0.3 else 0,
                                extra_biws=[sy.flex_extra_biw(rng.randint(8), rng.randint(1 << 14)) for _ in range(2)][:0])
        for _ in range(rng.randint(0, 12)):
            w[rng.randint(88)] ^= np.uint32(1 << rng.randint(32))
        for _ in range(rng.randint(0, 3)):
            w[rng.randint(88)] ^= np.uint32(rng.randint(1, 1 << 31))
        return w
    if kind == 1:        # random codewords everywhere behind a checksummed BIW
        w = np.array([sy.flex_codeword(int(v)) for v in rng.randint(0, 1 << 21, 88)], np.uint32)
        vsw = rng.randint(0, 40)
        w[0] = sy.flex_biw(vsw, eob=rng.randint(0, 4), priority=rng.randint(16))
        for i in range(vsw, min(88, vsw + 20)):          # vectors that pass their checksum, with arbitrary fields
            w[i] = sy.flex_codeword(sy.flex_checksummed(int(rng.randint(0, 1 << 21))))
        for i in range(1, 1 + rng.randint(0, 4)):
            w[i] = sy.flex_extra_biw(rng.randint(8), rng.randint(1 << 14)) if rng.rand() < 0.7 else w[i]
        return w
    if kind == 2:        # short addresses + checksummed vectors of every type, bodies anywhere (also past the end)
        w = np.array([sy.flex_codeword(int(v)) for v in rng.randint(0, 1 << 21, 88)], np.uint32)
        na = rng.randint(1, 12)
        w[0] = sy.flex_biw(1 + na)
        for i in range(na):
            w[1 + i] = sy.flex_codeword(int(rng.randint(0x8001, 0x1E0000)) if rng.rand() < 0.8 else int(rng.randint(1, 0x8000)))
            vec = rng.randint(8) << 4 | rng.randint(128) << 7 | rng.randint(128) << 14
            if rng.rand() < 0.5:
                vec = (vec & ~(0x7F << 7)) | rng.randint(1 + 2 * na, 80) << 7
            if rng.rand() < 0.5:
                vec = (vec & ~(0x7F << 14)) | rng.randint(0, 9) << 14
            w[1 + na + i] = sy.flex_codeword(sy.flex_checksummed(int(vec)))
        return w
    return rng.randint(0, 1 << 32, 88, dtype=np.uint64).astype(np.uint32)      # noise


def test_host_walk_matches_oracle_walk(ora, pkg):
    sy = pkg.synth
    rng = np.random.RandomState(2024)
    hp = HostFlex()
    kinds = set()
    for it in range(3000):
        w = _random_phase(sy, rng)
        coding, phase, cycle, frame = rng.randint(4), rng.randint(4), rng.randint(16), rng.randint(128)
        want_w, want = ora.flex_phase_process(w, coding, phase, cycle, frame)
        hp.out = []
        got_w = hp.phase(w, sy.FLEX_CODINGS[coding]["baud"], phase, cycle, frame)
        want = [m[:10] for m in want]
        assert hp.out == want, f"iteration {it}"
        assert np.array_equal(got_w, want_w), f"iteration {it}: in-place corrections differ"
        kinds |= {m[0] for m in want}
    hp.close()
    assert kinds >= {1, 2, 3, 16, 17, 18, 19, 20, 21, 22, 23, 24}, kinds   # every outcome of the walk was seen


def test_host_walk_long_messages_and_limits(ora, pkg):
    """the 255-character cut (pager_flex.c:664,769), numeric bodies of every length, zero-length alphanumerics"""
    sy = pkg.synth
    hp = HostFlex()
    cases = [[dict(kind="alnum", capcode=1000, text="x" * 200)], [dict(kind="alnum", capcode=1000, text="")],
             [dict(kind="alnum", long=(5, 0x1FFFFF), text="y" * 180)], [dict(kind="numeric", capcode=5, digits="1" * 40)]]
    cases += [[dict(kind="numeric", capcode=5 + n, digits="7" * n)] for n in range(1, 41, 3)]
    cases += [[dict(kind="numeric", long=(9, 0x1FFFF0 - n), digits="48" * n)] for n in range(1, 20, 2)]
    for recs in cases:
        w = sy.flex_phase_words(recs)
        want_w, want = ora.flex_phase_process(w, 0, 0, 1, 2)
        hp.out = []
        got_w = hp.phase(w, 1600, 0, 1, 2)
        assert hp.out == [m[:10] for m in want] and np.array_equal(got_w, want_w)
        assert _pages(want) == _expected(recs, 1600, 0, 1, 2)
    # an over-long alphanumeric: the vector's 7-bit length lets 127 words = 379 characters be announced
    w = sy.flex_phase_words([dict(kind="raw", capcode=77, vtype=5, body=[3 << 11] + [0x41 | 0x42 << 7 | 0x43 << 14] * 80)])
    want_w, want = ora.flex_phase_process(w, 0, 0, 1, 2)
    hp.out = []
    hp.phase(w, 1600, 0, 1, 2)
    assert hp.out == [m[:10] for m in want] and len(want) == 1 and len(want[0][9]) <= 255
    hp.close()


# ---- GPU parity -------------------------------------------------------------------------------------------------

EV_FIELDS = ("type", "sample", "sync_sample", "coding", "eye", "a", "b", "inv_a", "fiw_raw", "fiw", "fiw_rc", "sample_range",
             "sample_delta", "cycle", "frame")


def _ev_tuple(e):
    return tuple(int(e[k]) for k in EV_FIELDS)


def _check_channel(ora, sy, got_ev, got_fw, want_ev, tag):
    assert [_ev_tuple(e) for e in got_ev] == [_ev_tuple(e) for e in want_ev], tag
    for g, w in zip(got_ev, want_ev):
        if int(g["type"]) == 1:
            c = sy.FLEX_CODINGS[int(g["coding"])]
            assert int(g["baud"]) == c["baud"] and int(g["nr_phases"]) == len(c["phases"])
            assert np.array_equal(got_fw[int(g["frame_index"])]["words"], w["words"]), tag


def _flex_channels(sy, n_frames=2, seed=0):
    """nine channels: the four codings clean and noisy, damaged sync words, a long 800 Hz tone in front of a frame
    (a BS1 run far longer than 256 samples), DC offsets, silence and noise"""
    rng = np.random.RandomState(seed)
    chans = []

    def frames(coding, k, **kw):
        out = []
        for i in range(k):
            ph = {p: sy.flex_phase_words([dict(r) for r in RECORDS[: 3 + (i + p) % 5]]) for p in sy.FLEX_CODINGS[coding]["phases"]}
            out.append(sy.flex_frame_levels(coding, (3 + i) % 16, (7 * coding + i) % 128, ph, **kw))
        return out

    chans.append(sy.flex_pcm(frames(0, n_frames), lead=333, trail=900, noise=300, seed=1))
    chans.append(sy.flex_pcm(frames(1, n_frames), lead=1, trail=900, noise=1500, seed=2))
    chans.append(sy.flex_pcm(frames(2, n_frames), lead=4099, trail=900, noise=500, seed=3, offset=700))
    chans.append(sy.flex_pcm(frames(3, n_frames), lead=77, trail=900, noise=500, seed=4, offset=-400, amplitude=5000))
    bad = [sy.flex_frame_levels(1, 1, 3, {}, a_flip=0x0F0F0000), sy.flex_frame_levels(2, 1, 4, {}, fiw_flip=0x00700000)]
    chans.append(sy.flex_pcm(bad + frames(3, 1), lead=10, trail=900, noise=200, seed=5))
    tone = [(3 if (k & 1) == 0 else -3, 10) for k in range(700)]            # 7000 samples of 1010..: one long run
    chans.append(sy.flex_pcm([tone + frames(0, 1)[0]] + frames(2, 1), lead=50, trail=900, noise=100, seed=6))
    alt = sy.flex_frame_levels(0, 2, 9, {0: sy.flex_phase_words([], idle=(0, 0x1FFFFF))})   # idle fill that reads 1010..
    chans.append(sy.flex_pcm([alt] + frames(0, 1), lead=5, trail=900, noise=400, seed=7))
    chans.append(rng.randint(-20000, 20000, 1000).astype(np.int16))          # noise only
    chans.append(np.zeros(1000, np.int16))                                   # silence: every bit a one, no swing
    n = max(len(c) for c in chans)
    return np.stack([np.concatenate([c, rng.randint(-300, 300, n - len(c)).astype(np.int16)]) for c in chans])


@pytest.mark.gpu
def test_gpu_flex_events_match_oracle(ora, pkg):
    sy = pkg.synth
    pcm = _flex_channels(sy)
    C_, n = pcm.shape
    fx = pkg.binding.Flex(C_, n)
    assert pkg.binding.FLEX_EVENT_DTYPE.itemsize == 88 and pkg.binding.FLEX_FRAME_DTYPE.itemsize == 4 * 88 * 4
    ev, fw = fx.process_host(pcm)
    total = 0
    for c in range(C_):
        want_ev, _ = ora.Flex().feed(pcm[c])
        _check_channel(ora, sy, ev[ev["channel"] == c], fw, want_ev, f"channel {c}")
        total += len(want_ev)
    assert total >= 14 and {int(t) for t in ev["type"]} == {1, 2, 3}
    # nothing new: no events; the object keeps its place in the stream
    ev2, fw2 = fx.process_host(np.zeros((C_, 0), np.int16))
    assert ev2.size == 0 and fw2.size == 0
    fx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("chunks", [[30000] * 9, [1, 309, 311, 5000, 1117, 30001, 7, 64000, 9973, 33333, 10 ** 9], [4096] * 80])
def test_gpu_flex_streaming(ora, pkg, chunks):
    """frames that straddle calls: sync words, the wait for the FIW and the block gather all reach back into the
    history ring"""
    sy = pkg.synth
    pcm = _flex_channels(sy, n_frames=3, seed=1)
    C_, n = pcm.shape
    fx = pkg.binding.Flex(C_, 64000)
    got = [[] for _ in range(C_)]
    pos = 0
    for k in chunks:
        k = min(k, 64000)
        blk = pcm[:, pos:pos + k]
        if blk.shape[1] == 0:
            break
        ev, fw = fx.process_host(blk)
        for e in ev:
            got[int(e["channel"])].append((e, fw[int(e["frame_index"])]["words"].copy() if int(e["type"]) == 1 else None))
        pos += blk.shape[1]
    for c in range(C_):
        want_ev, _ = ora.Flex().feed(pcm[c, :pos])
        assert [_ev_tuple(e) for e, _ in got[c]] == [_ev_tuple(e) for e in want_ev], f"channel {c}"
        for (e, w), we in zip(got[c], want_ev):
            if w is not None:
                assert np.array_equal(w, we["words"]), f"channel {c}"
    fx.close()


@pytest.mark.gpu
def test_gpu_flex_pages_end_to_end(ora, pkg):
    """GPU stage -> C host walk -> callbacks == the oracle's messages, channel by channel"""
    sy = pkg.synth
    pcm = _flex_channels(sy, n_frames=2, seed=2)
    C_, n = pcm.shape
    fx = pkg.binding.Flex(C_, n)
    ev, fw = fx.process_host(pcm)
    pages = 0
    for c in range(C_):
        _, want = ora.Flex().feed(pcm[c])
        hp = HostFlex()
        hp.on_events(ev[ev["channel"] == c], fw)
        hp.close()
        assert hp.out == [m[:10] for m in want], f"channel {c}"
        pages += len(_pages(want))
    assert pages > 40
    fx.close()


@pytest.mark.gpu
def test_gpu_flex_many_channels_and_strides(ora, pkg):
    """more channels than compute units, each with its own lead-in; input rows wider than the block (in_stride)"""
    sy = pkg.synth
    rng = np.random.RandomState(9)
    base = [sy.flex_frame_levels(k, k, 10 + k, {p: sy.flex_phase_words(RECORDS[:4]) for p in sy.FLEX_CODINGS[k]["phases"]})
            for k in range(4)]
    C_, n = 300, 33000
    pcm = np.zeros((C_, n), np.int16)
    for c in range(C_):
        x = sy.flex_pcm([base[c % 4]], lead=int(rng.randint(0, 2500)), trail=0, noise=float(rng.randint(50, 2500)), seed=c)
        pcm[c, :min(n, x.size)] = x[:n]
    fx = pkg.binding.Flex(C_, n)
    ev, fw = fx.process_host(pcm)
    nframes = 0
    for c in range(0, C_, 7):
        want_ev, _ = ora.Flex().feed(pcm[c])
        _check_channel(ora, sy, ev[ev["channel"] == c], fw, want_ev, f"channel {c}")
        nframes += sum(int(e["type"]) == 1 for e in want_ev)
    assert nframes >= 35
    assert int((ev["type"] == 1).sum()) >= 280
    fx.close()
    with pytest.raises(pkg.binding.MfmError):
        pkg.binding.Flex(0, 100)


@pytest.mark.gpu
def test_gpu_flex_full_block_of_busy_channels(ora, pkg):
    """the block the headline chain hands over (64 channels x 447 392 samples = one 2^26-sample IQ block at 16 kHz), every
    channel full of back-to-back frames of the four codings at its own offset: about 900 frames in one call, then the same
    stream again in two calls - events and words against the oracle, channel by channel"""
    sy = pkg.synth
    C_, n = 64, 447392
    recs = [dict(kind="alnum", capcode=1000 + i, text="THE QUICK BROWN FOX JUMPS OVER THE LAZY DOG %d" % i) for i in range(4)]
    frames = []
    for k in range(4):
        ph = {p: sy.flex_phase_words(recs) for p in sy.FLEX_CODINGS[k]["phases"]}
        frames.append(sy.flex_pcm([sy.flex_frame_levels(k, 1, k, ph)], noise=300, seed=k))
    pcm = np.stack([np.concatenate([frames[c % 4]] * (n // 30000 + 2))[(c * 977) % 30000:][:n] for c in range(C_)])
    fx = pkg.binding.Flex(C_, n)
    ev, fw = fx.process_host(pcm)
    ev2, fw2 = fx.process_host(pcm[:, :200001])
    ev3, fw3 = fx.process_host(pcm[:, 200001:])
    fx.close()
    frames_seen = 0
    for c in range(C_):
        o = ora.Flex()
        w1, _ = o.feed(pcm[c])
        w2, _ = o.feed(pcm[c, :200001])
        w3, _ = o.feed(pcm[c, 200001:])
        _check_channel(ora, sy, ev[ev["channel"] == c], fw, w1, f"channel {c}")
        _check_channel(ora, sy, ev2[ev2["channel"] == c], fw2, w2, f"channel {c}, second pass, first call")
        _check_channel(ora, sy, ev3[ev3["channel"] == c], fw3, w3, f"channel {c}, second pass, second call")
        frames_seen += sum(int(e["type"]) == 1 for e in w1)
    assert frames_seen >= 850


@pytest.mark.gpu
def test_gpu_flex_event_list_limits(ora, pkg):
    """a caller-chosen max_events that is too small is reported (MFM_E_STATE), never a silent loss or an overrun; bad
    arguments are refused"""
    sy = pkg.synth
    tone = [(3 if (k & 1) == 0 else -3, 10) for k in range(3000)]   # 800 Hz: an unknown-mode event every 1120 samples
    pcm = sy.flex_pcm([tone], lead=5, trail=100, noise=300, seed=1)[None, :]
    fx = pkg.binding.Flex(1, pcm.shape[1], max_events=3)
    with pytest.raises(pkg.MfmError) as ei:
        fx.process_host(pcm)
    assert ei.value.code == pkg.binding.MFM_E_STATE
    fx.close()
    fx = pkg.binding.Flex(1, pcm.shape[1])
    ev, fw = fx.process_host(pcm)
    want, _ = ora.Flex().feed(pcm[0])
    assert len(ev) == len(want) > 10
    with pytest.raises(pkg.MfmError):
        fx.process_host(np.zeros((1, pcm.shape[1] + 1), np.int16))     # more than max_in_samples
    fx.close()
    with pytest.raises(pkg.MfmError):
        pkg.binding.Flex(1, (1 << 26) + 1)


# ---- the decoder-shaped driver ---------------------------------------------------------------------------------

def _esc(text):
    """decoder.c:121-166"""
    out = ""
    for ch in text:
        c = chr(ch)
        if c in "\n\r":
            out += "\\n"
        elif c == '"':
            out += '\\"'
        elif c == "\\":
            out += "\\\\"
        elif c == "/":
            out += "\\/"
        elif c == "\b":
            out += "<BKSP>"
        elif c == "\f":
            out += "<FF>"
        elif c == "\t":
            out += "\\t"
        elif ch in (3, 4, 0x17):
            out += " "
        elif 32 <= ch < 127:
            out += c
        else:
            out += "\\u%04x" % ch
    return out


def _flex_json_lines(msgs):
    """decoder.c:173-262 with the timestamp of MFM_DECODER_FIXED_TIME"""
    out = []
    for kind, baud, phase, cycle, frame, a0, a1, _a2, cap, text in [m[:10] for m in msgs]:
        head = ('{"proto":"flex","type":"%s","timestamp":"1970-01-01 00:00:00 UTC","baud":%d,"syncLevel":0,"frameNo":%d,'
                '"cycleNo":%d,"phaseNo":"%s","capCode":%d,')
        if kind == 1:
            out.append(head % ("alphanumeric", baud, frame, cycle, "ABCD"[phase], cap) +
                       '"fragment":%s,"maildrop":%s,"fragSeq":%d,"message":"%s"}\n' %
                       ("true" if a0 & 1 else "false", "true" if a0 & 2 else "false", a0 >> 2, _esc(text)))
        elif kind == 2:
            out.append(head % ("numeric", baud, frame, cycle, "ABCD"[phase], cap) + '"message":"%s"}\n' % _esc(text))
        elif kind == 3 and a0 == 0:
            out.append(head % ("tempAddrActivation", baud, frame, cycle, "ABCD"[phase], cap) +
                       '"startFrameNo":%d,"tempAddressId":%d}\n' % (a1 & 0x7F, (a1 >> 7) & 0xF))
    return "".join(out)


@pytest.mark.gpu
@pytest.mark.parametrize("opts", [[], ["-i", "-b", "-p", "0.9999"]])
def test_decoder_amd_flex_json_matches_oracle(tmp_path, ora, pkg, opts):
    """decoder/decoder.c's FLEX path (:580-673 loop, :173-262 output; FLEX is its default protocol): four 25 kS/s PCM
    files, one per coding -> 16/25 -> 16 000 Hz -> FLEX; the JSON lines must be what the oracle chain's messages print as."""
    import json
    import subprocess
    sy = pkg.synth
    tool = os.path.join(os.path.dirname(HOST_SO), "decoder_amd")
    recs = RECORDS[:8] + [dict(kind="alnum", capcode=31, text='quote " slash / back \\ tab\t nl\n bell\x07 end\x17')]
    chans = []
    for k in range(4):
        fr = [sy.flex_frame_levels(k, 2 + i, 9 * k + i, {p: sy.flex_phase_words(recs[i:] + recs[:i]) for p in sy.FLEX_CODINGS[k]["phases"]})
              for i in range(2)]
        chans.append(sy.flex_pcm(fr, lead=3000 + 517 * k, trail=4000, noise=300, seed=k, rate=25000, offset=500 if "-b" in opts else 0))
    total = min(len(x) for x in chans)
    taps = sy.design_lpf(321, 0.45 / 25, 1.0) * 16
    (tmp_path / "filter.json").write_text(json.dumps({"lpfCoeffs": [float(t) for t in taps]}))
    invert = "-i" in opts
    paths = []
    for c, x in enumerate(chans):
        p = tmp_path / f"ch{c}.pcm"
        p.write_bytes(((-x[:total].astype(np.int32)).astype(np.int16) if invert else x[:total]).tobytes())
        paths.append(str(p))
    out = tmp_path / "pages.json"
    env = dict(os.environ, MFM_DECODER_FIXED_TIME="1")
    r = subprocess.run([tool, "-I", "16", "-D", "25", "-S", "25000", "-F", str(tmp_path / "filter.json"), "-f", "929612500",
                        "-c", "-o", str(out), "-B", "30000"] + opts + paths, capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    rtaps = ora.quantize_taps(taps)
    for c, x in enumerate(chans):
        xin = (-x[:total].astype(np.int32)).astype(np.int16) if invert else x[:total]
        res = ora.Resampler(rtaps, 16, 25, dc_pole=0.9999 if "-b" in opts else None, invert=invert)
        _, msgs = ora.Flex().feed(res.feed(xin))
        assert len(_pages(msgs)) >= 14 * len(sy.FLEX_CODINGS[c]["phases"]), f"channel {c}: the synthetic pages did not decode"
        assert (tmp_path / f"pages.json.{c}").read_text() == _flex_json_lines(msgs), f"channel {c}"


@pytest.mark.gpu
def test_flex_chain_through_both_binaries(tmp_path, ora, pkg):
    """The reference's FLEX deployment end to end at its process boundary: a cs16 capture at 2.4 MS/s with four FLEX
    carriers (one per coding, 2- and 4-level FSK) -> multifm_amd (D = 96, the 128-tap 25 kHz low-pass: the headline
    geometry) -> one 25 kS/s PCM sink per channel -> decoder_amd (16/25 -> 16 000 Hz -> FLEX) -> JSON lines.  Must
    equal the oracle chain run on the same bytes."""
    import json
    import subprocess
    sy = pkg.synth
    fs, decim = 2400000, 96
    center = 929612500
    offs = [-600000, -137500, 212500, 875000]
    n16 = 33500
    acc = np.zeros((n16 * fs // 16000, 2), np.float64)
    for k, o in enumerate(offs):
        ph = {p: sy.flex_phase_words(RECORDS[k:k + 5]) for p in sy.FLEX_CODINGS[k]["phases"]}
        iq = sy.flex_fm_iq([sy.flex_frame_levels(k, 7, 30 + k, ph)], fs, float(o), amplitude=5000.0, lead=800 + 300 * k, trail=2700 - 300 * k,
                           noise=60.0, seed=k)
        acc[:iq.shape[0]] += iq[:acc.shape[0]]
    raw = np.clip(np.round(acc), -32768, 32767).astype(np.int16)
    cap = tmp_path / "capture.cs16"
    cap.write_bytes(raw.tobytes())
    taps_file = os.path.join(ROOT, "etc", "lpf_25khz_2400k_128.json")
    lpf = np.array(json.load(open(taps_file))["lpfTaps"])
    cfg = {"device": {"type": "file", "filename": str(cap), "fileFormat": "cs16"}, "sampleRateHz": fs, "centerFreqHz": center,
           "nrSampBufs": 32, "decimationFactor": decim,
           "channels": [{"outFifo": str(tmp_path / f"ch{c}.pcm"), "chanCenterFreq": int(center + o)} for c, o in enumerate(offs)]}
    for c in range(4):
        (tmp_path / f"ch{c}.pcm").write_bytes(b"")
    (tmp_path / "cfg.json").write_text(json.dumps(cfg))
    host_dir = os.path.dirname(HOST_SO)
    r = subprocess.run([os.path.join(host_dir, "multifm_amd"), str(tmp_path / "cfg.json"), taps_file], capture_output=True, text=True,
                       timeout=180)
    assert r.returncode == 0, r.stderr[-2000:]
    rt = sy.design_lpf(321, 0.45 / 25, 1.0) * 16
    (tmp_path / "filter.json").write_text(json.dumps({"lpfCoeffs": [float(t) for t in rt]}))
    env = dict(os.environ, MFM_DECODER_FIXED_TIME="1")
    r = subprocess.run([os.path.join(host_dir, "decoder_amd"), "-I", "16", "-D", "25", "-S", "25000", "-F", str(tmp_path / "filter.json"),
                        "-f", str(center), "-m", "FLEX", "-c", "-o", str(tmp_path / "pages.json")] +
                       [str(tmp_path / f"ch{c}.pcm") for c in range(4)], capture_output=True, text=True, timeout=180, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    iq = raw.reshape(-1, 2)
    cre = np.stack([ora.make_taps(lpf, int(o), fs, 1.0)[0] for o in offs])
    cim = np.stack([ora.make_taps(lpf, int(o), fs, 1.0)[1] for o in offs])
    incr = np.stack([ora.rot_incr(int(o), fs, decim) for o in offs])
    pcm, _ = ora.run_channels(iq, cre, cim, incr, decim, threads=4)
    rtaps = ora.quantize_taps(rt)
    for c in range(4):
        got_pcm = np.frombuffer((tmp_path / f"ch{c}.pcm").read_bytes(), dtype=np.int16)
        assert np.array_equal(got_pcm, pcm[c]), f"channel {c}: PCM sink differs"
        _, msgs = ora.Flex().feed(ora.Resampler(rtaps, 16, 25).feed(pcm[c]))
        assert len(_pages(msgs)) >= 3 * len(sy.FLEX_CODINGS[c]["phases"]), f"channel {c}: the synthetic pages did not decode"
        assert (tmp_path / f"pages.json.{c}").read_text() == _flex_json_lines(msgs), f"channel {c}: JSON lines differ"
