"""ctypes wrapper of oracle/liboracle.so and oracle/_ref (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "liboracle.so")
REF_ATAN2_SO = os.path.join(ROOT, "oracle", "_ref", "libref_fast_atan2f.so")

_i16p = C.POINTER(C.c_int16)
_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_SO):
            raise RuntimeError(f"{ORACLE_SO} missing: run `make -C oracle`")
        o = C.CDLL(ORACLE_SO)
        o.mfmo_r14.argtypes = [C.c_int32]
        o.mfmo_r14.restype = C.c_int16
        o.mfmo_atan_table.argtypes = [C.POINTER(C.c_float)]
        o.mfmo_fast_atan2f.argtypes = [C.c_float, C.c_float]
        o.mfmo_fast_atan2f.restype = C.c_float
        o.mfmo_fast_atan2f_fma.argtypes = [C.c_float, C.c_float]
        o.mfmo_fast_atan2f_fma.restype = C.c_float
        o.mfmo_phi_to_pcm.argtypes = [C.c_float]
        o.mfmo_phi_to_pcm.restype = C.c_int16
        o.mfmo_phi_to_pcm_range.argtypes = [C.c_uint32, C.c_uint32, _i16p]
        o.mfmo_phi_to_pcm_range.restype = None
        o.mfmo_discriminate_batch.argtypes = [C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_size_t, _i16p, C.c_int]
        o.mfmo_discriminate_batch.restype = None
        o.mfmo_fm_step.argtypes = [C.c_int16, C.c_int16, C.c_int32, C.c_int32]
        o.mfmo_fm_step.restype = C.c_int16
        o.mfmo_make_taps.argtypes = [C.POINTER(C.c_double), C.c_size_t, C.c_int32, C.c_uint32, C.c_double, _i16p, _i16p]
        o.mfmo_make_taps.restype = None
        o.mfmo_rot_incr.argtypes = [C.c_int32, C.c_uint32, C.c_uint, _i16p, _i16p]
        o.mfmo_rot_incr.restype = None
        o.mfmo_gain_from_db.argtypes = [C.c_double]
        o.mfmo_gain_from_db.restype = C.c_double
        o.mfmo_rot_step.argtypes = [_i16p, _i16p, C.c_int16, C.c_int16]
        o.mfmo_rot_step.restype = None
        o.mfmo_chan_new.argtypes = [_i16p, _i16p, C.c_size_t, C.c_uint, C.c_int16, C.c_int16]
        o.mfmo_chan_new.restype = C.c_void_p
        o.mfmo_chan_free.argtypes = [C.c_void_p]
        o.mfmo_chan_free.restype = None
        o.mfmo_chan_feed.argtypes = [C.c_void_p, _i16p, C.c_size_t, _i16p, _i16p, C.c_size_t]
        o.mfmo_chan_feed.restype = C.c_size_t
        o.mfmo_chan_skip_outputs.argtypes = [C.c_void_p, C.c_uint64]
        o.mfmo_chan_skip_outputs.restype = None
        o.mfmo_chan_rot.argtypes = [C.c_void_p, _i16p, _i16p]
        o.mfmo_chan_rot.restype = None
        o.mfmo_run_channels.argtypes = [_i16p, C.c_size_t, C.c_size_t, _i16p, _i16p, C.c_size_t, C.c_uint, _i16p,
                                        _i16p, _i16p, C.c_size_t, C.c_uint]
        o.mfmo_run_channels.restype = C.c_size_t
        o.mfmo_twoslot_run.argtypes = [_i16p, C.c_size_t, C.c_size_t, _i16p, _i16p, C.c_size_t, C.c_uint, C.c_int16,
                                       C.c_int16, _i16p, _i16p, C.c_size_t]
        o.mfmo_twoslot_run.restype = C.c_size_t
        o.mfmo_resampler_new.argtypes = [_i16p, C.c_size_t, C.c_uint, C.c_uint]
        o.mfmo_resampler_new.restype = C.c_void_p
        o.mfmo_resampler_free.argtypes = [C.c_void_p]
        o.mfmo_resampler_free.restype = None
        o.mfmo_resampler_phase_len.argtypes = [C.c_void_p]
        o.mfmo_resampler_phase_len.restype = C.c_size_t
        o.mfmo_resampler_feed.argtypes = [C.c_void_p, _i16p, C.c_size_t, _i16p, C.c_size_t]
        o.mfmo_resampler_feed.restype = C.c_size_t
        o.mfmo_dc_blocker_init.argtypes = [C.c_void_p, C.c_double]
        o.mfmo_dc_blocker_init.restype = None
        o.mfmo_dc_blocker_apply.argtypes = [C.c_void_p, _i16p, C.c_size_t]
        o.mfmo_dc_blocker_apply.restype = None
        o.mfmo_resampler_quantize_taps.argtypes = [C.POINTER(C.c_double), C.c_size_t, _i16p]
        o.mfmo_resampler_quantize_taps.restype = None
        o.mfmo_unpack_bytes.argtypes = [C.c_void_p, C.c_size_t, C.c_int, _i16p]
        o.mfmo_unpack_bytes.restype = None
        u32p = C.POINTER(C.c_uint32)
        o.mfmo_bch_tables.argtypes = [C.POINTER(C.c_int), C.POINTER(C.c_int)]
        o.mfmo_bch_tables.restype = None
        o.mfmo_bch3121_decode.argtypes = [u32p]
        o.mfmo_bch3121_decode_batch.argtypes = [u32p, C.POINTER(C.c_uint8), C.c_size_t, C.c_uint]
        o.mfmo_bch3121_decode_batch.restype = None
        o.mfmo_pocsag_new.restype = C.c_void_p
        o.mfmo_pocsag_free.argtypes = [C.c_void_p]
        o.mfmo_pocsag_free.restype = None
        o.mfmo_pocsag_on_pcm.argtypes = [C.c_void_p, _i16p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t),
                                         C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        o.mfmo_pocsag_msgdec_new.restype = C.c_void_p
        o.mfmo_pocsag_msgdec_free.argtypes = [C.c_void_p]
        o.mfmo_pocsag_msgdec_free.restype = None
        o.mfmo_pocsag_msgdec_batch.argtypes = [C.c_void_p, u32p, C.c_int, C.c_uint32, C.c_uint64, C.c_void_p, C.c_size_t,
                                               C.POINTER(C.c_size_t)]
        o.mfmo_flex_new.restype = C.c_void_p
        o.mfmo_flex_free.argtypes = [C.c_void_p]
        o.mfmo_flex_free.restype = None
        o.mfmo_flex_on_pcm.argtypes = [C.c_void_p, _i16p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t),
                                       C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        o.mfmo_flex_phase_process.argtypes = [u32p, C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_uint64, C.c_void_p,
                                              C.c_size_t, C.POINTER(C.c_size_t)]
        o.mfmo_mm_init.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float]
        o.mfmo_mm_init.restype = None
        o.mfmo_mm_process.argtypes = [C.c_void_p, _i16p, C.c_size_t, _i16p, C.c_size_t]
        o.mfmo_mm_process.restype = C.c_size_t
        f32p, f64p = C.POINTER(C.c_float), C.POINTER(C.c_double)
        o.mfmo_f32_chan_new.argtypes = [C.c_int32, C.c_uint32, C.c_uint32, f64p, C.c_size_t, C.c_double]
        o.mfmo_f32_chan_new.restype = C.c_void_p
        o.mfmo_f32_chan_free.argtypes = [C.c_void_p]
        o.mfmo_f32_chan_free.restype = None
        o.mfmo_f32_chan_taps.argtypes = [C.c_void_p, f64p, f64p]
        o.mfmo_f32_chan_taps.restype = None
        o.mfmo_f32_chan_push.argtypes = [C.c_void_p, f32p, C.c_size_t, f64p, f64p, C.c_size_t]
        o.mfmo_f32_chan_push.restype = C.c_size_t
        _lib = o
    return _lib


def ref_atan2():
    """The reference's own fast_atan2f object code (oracle/_ref), or None if it was not built."""
    global _ref
    if _ref is None and os.path.exists(REF_ATAN2_SO):
        r = C.CDLL(REF_ATAN2_SO)
        r.fast_atan2f.argtypes = [C.c_float, C.c_float]
        r.fast_atan2f.restype = C.c_float
        _ref = r
    return _ref


def p16(a):
    return a.ctypes.data_as(_i16p)


def make_taps(lpf_taps, offset_hz, sample_rate, gain=1.0):
    h = np.ascontiguousarray(lpf_taps, dtype=np.float64)
    cre = np.zeros(h.size, np.int16)
    cim = np.zeros(h.size, np.int16)
    lib().mfmo_make_taps(h.ctypes.data_as(C.POINTER(C.c_double)), h.size, int(offset_hz), int(sample_rate),
                         float(gain), p16(cre), p16(cim))
    return cre, cim


def rot_incr(offset_hz, sample_rate, decimation):
    a = np.zeros(2, np.int16)
    lib().mfmo_rot_incr(int(offset_hz), int(sample_rate), int(decimation), p16(a[0:1]), p16(a[1:2]))
    return a


def atan_table():
    t = (C.c_float * 257)()
    lib().mfmo_atan_table(t)
    return np.frombuffer(t, dtype=np.float32).copy()


def expected_outputs(nr_samples, nr_taps, decimation):
    return 0 if nr_samples < nr_taps else (nr_samples - nr_taps) // decimation + 1


def run_channels(iq, cre, cim, incr, decimation, threads=1, want_iq=False):
    """Whole-buffer oracle run.  iq: (n,2) int16; cre/cim: (C,T); incr: (C,2). Returns (pcm[C][N], iq[C][N][2]|None)."""
    iq = np.ascontiguousarray(iq, dtype=np.int16)
    cre = np.ascontiguousarray(cre, dtype=np.int16)
    cim = np.ascontiguousarray(cim, dtype=np.int16)
    incr = np.ascontiguousarray(incr, dtype=np.int16)
    nch, T = cre.shape
    n = iq.shape[0]
    nout = expected_outputs(n, T, decimation)
    pcm = np.zeros((nch, max(nout, 1)), np.int16)
    iqo = np.zeros((nch, max(nout, 1), 2), np.int16) if want_iq else None
    got = lib().mfmo_run_channels(p16(iq), n, nch, p16(cre), p16(cim), T, decimation, p16(incr), p16(pcm),
                                  p16(iqo) if want_iq else None, max(nout, 1), threads)
    assert got == nout, (got, nout)
    return pcm[:, :nout], (iqo[:, :nout] if want_iq else None)


class Channel:
    """Streaming oracle channel (arbitrary chunking)."""

    def __init__(self, cre, cim, decimation, incr):
        self.cre = np.ascontiguousarray(cre, dtype=np.int16)
        self.cim = np.ascontiguousarray(cim, dtype=np.int16)
        self.h = lib().mfmo_chan_new(p16(self.cre), p16(self.cim), self.cre.size, decimation, int(incr[0]), int(incr[1]))
        if not self.h:
            raise ValueError("mfmo_chan_new rejected the configuration")
        self.decim = decimation

    def feed(self, iq):
        iq = np.ascontiguousarray(iq, dtype=np.int16).reshape(-1, 2)
        cap = iq.shape[0] // self.decim + 2 + self.cre.size
        pcm = np.zeros(cap, np.int16)
        q = np.zeros((cap, 2), np.int16)
        n = lib().mfmo_chan_feed(self.h, p16(iq), iq.shape[0], p16(pcm), p16(q), cap)
        return pcm[:n], q[:n]

    def skip_outputs(self, n):
        """advance the rotator as if n outputs had been produced (window checks in the middle of a long stream)"""
        lib().mfmo_chan_skip_outputs(self.h, int(n))

    def rot(self):
        a = np.zeros(2, np.int16)
        lib().mfmo_chan_rot(self.h, p16(a[0:1]), p16(a[1:2]))
        return a

    def close(self):
        if self.h:
            lib().mfmo_chan_free(self.h)
            self.h = None

    def __del__(self):
        self.close()


def twoslot_run(iq, buf_samples, cre, cim, decimation, incr):
    iq = np.ascontiguousarray(iq, dtype=np.int16).reshape(-1, 2)
    nb = iq.shape[0] // buf_samples
    cre = np.ascontiguousarray(cre, dtype=np.int16)
    cim = np.ascontiguousarray(cim, dtype=np.int16)
    cap = iq.shape[0] // decimation + 2
    pcm = np.zeros(cap, np.int16)
    q = np.zeros((cap, 2), np.int16)
    n = lib().mfmo_twoslot_run(p16(iq), buf_samples, nb, p16(cre), p16(cim), cre.size, decimation, int(incr[0]),
                               int(incr[1]), p16(pcm), p16(q), cap)
    return pcm[:n], q[:n]


class Resampler:
    """Oracle rational resampler (one channel), optional DC blocker and input inversion, arbitrary chunking."""

    def __init__(self, coeffs_q14, interpolate, decimate, dc_pole=None, invert=False):
        self.co = np.ascontiguousarray(coeffs_q14, dtype=np.int16)
        self.h = lib().mfmo_resampler_new(p16(self.co), self.co.size, interpolate, decimate)
        assert self.h
        self.interp, self.decim, self.invert = interpolate, decimate, invert
        self.dc = None
        if dc_pole is not None:
            self.dc = (C.c_int32 * 4)()
            lib().mfmo_dc_blocker_init(self.dc, float(dc_pole))

    def phase_len(self):
        return lib().mfmo_resampler_phase_len(self.h)

    def feed(self, pcm):
        x = np.ascontiguousarray(pcm, dtype=np.int16).copy()
        if self.invert:
            x = (-x.astype(np.int32)).astype(np.int16)  # decoder.c:624 on int16 storage
        cap = x.size * self.interp // self.decim + 8
        out = np.zeros(cap, np.int16)
        n = lib().mfmo_resampler_feed(self.h, p16(x), x.size, p16(out), cap)
        out = out[:n].copy()
        if self.dc is not None and n:
            lib().mfmo_dc_blocker_apply(self.dc, p16(out), n)
        return out

    def close(self):
        if self.h:
            lib().mfmo_resampler_free(self.h)
            self.h = None

    def __del__(self):
        self.close()


def quantize_taps(taps):
    t = np.ascontiguousarray(taps, dtype=np.float64)
    out = np.zeros(t.size, np.int16)
    lib().mfmo_resampler_quantize_taps(t.ctypes.data_as(C.POINTER(C.c_double)), t.size, p16(out))
    return out


# ---- BCH(31,21) and POCSAG (oracle/pocsag_oracle.c) --------------------------------------------------------

# struct mfmo_pocsag_event / struct mfmo_pocsag_msg
POCSAG_EVENT_DTYPE = np.dtype([("type", "<u4"), ("baud", "<u4"), ("sample", "<u8"), ("aux", "<u4"), ("nr_ok", "<u4"),
                               ("fail_mask", "<u4"), ("pad", "<u4"), ("raw", "<u4", (16,)), ("corrected", "<u4", (16,))])
POCSAG_MSG_DTYPE = np.dtype([("type", "<u4"), ("baud", "<u4"), ("capcode", "<u4"), ("function", "<u4"), ("len", "<u4"),
                             ("pad", "<u4"), ("sample", "<u8"), ("text", "S512")])
EV_SYNC_FOUND, EV_BATCH, EV_SYNC_LOST, EV_SYNC_KEPT = 1, 2, 3, 4


def bch_tables():
    a, i = (C.c_int * 32)(), (C.c_int * 32)()
    lib().mfmo_bch_tables(a, i)
    return list(a)[:31], list(i)


def bch3121_decode(word):
    v = C.c_uint32(int(word))
    rc = lib().mfmo_bch3121_decode(C.byref(v))
    return rc, v.value


def bch3121_decode_batch(words, threads=1):
    w = np.ascontiguousarray(words, dtype=np.uint32).copy()
    rc = np.zeros(w.size, np.uint8)
    lib().mfmo_bch3121_decode_batch(w.ctypes.data_as(C.POINTER(C.c_uint32)), rc.ctypes.data_as(C.POINTER(C.c_uint8)),
                                    w.size, threads)
    return w, rc


def _msg_tuple(m):
    n = int(m["len"])
    raw = bytes(m["text"]).ljust(512, b"\0")[:min(n, 511)]  # numpy 'S' strips trailing NULs; put them back
    return int(m["type"]), int(m["baud"]), int(m["capcode"]), int(m["function"]), raw, int(m["sample"])


class Pocsag:
    """Oracle POCSAG decoder for one channel: feed(pcm) -> (events, messages) of that call."""

    def __init__(self):
        self.h = lib().mfmo_pocsag_new()
        assert self.h

    def feed(self, pcm):
        x = np.ascontiguousarray(pcm, dtype=np.int16)
        cap = x.size // 512 + 16
        ev = np.zeros(cap, POCSAG_EVENT_DTYPE)
        ms = np.zeros(cap, POCSAG_MSG_DTYPE)
        nev, nms = C.c_size_t(0), C.c_size_t(0)
        if x.size:
            lib().mfmo_pocsag_on_pcm(self.h, p16(x), x.size, ev.ctypes.data, cap, C.byref(nev), ms.ctypes.data, cap,
                                     C.byref(nms))
        assert nev.value <= cap and nms.value <= cap
        return ev[:nev.value].copy(), [_msg_tuple(m) for m in ms[:nms.value]]

    def close(self):
        if self.h:
            lib().mfmo_pocsag_free(self.h)
            self.h = None

    def __del__(self):
        self.close()


class PocsagMsgDec:
    """Oracle message layer alone (_process_batch + deliver), driven by batches collected elsewhere."""

    def __init__(self):
        self.h = lib().mfmo_pocsag_msgdec_new()
        assert self.h

    def _call(self, words, flush, baud, sample):
        ms = np.zeros(40, POCSAG_MSG_DTYPE)
        n = C.c_size_t(0)
        wp = None
        if words is not None:
            w = np.ascontiguousarray(words, dtype=np.uint32)
            wp = w.ctypes.data_as(C.POINTER(C.c_uint32))
        r = lib().mfmo_pocsag_msgdec_batch(self.h, wp, flush, baud, sample, ms.ctypes.data, 40, C.byref(n))
        return r, [_msg_tuple(m) for m in ms[:n.value]]

    def batch(self, words, baud, sample):
        return self._call(words, 0, baud, sample)

    def flush(self, baud, sample):
        return self._call(None, 1, baud, sample)[1]

    def close(self):
        if self.h:
            lib().mfmo_pocsag_msgdec_free(self.h)
            self.h = None

    def __del__(self):
        self.close()


# ---- FLEX (oracle/flex_oracle.c) ---------------------------------------------------------------------------

# struct mfmo_flex_event / struct mfmo_flex_msg
FLEX_EVENT_DTYPE = np.dtype([("type", "<u4"), ("coding", "<u4"), ("sample", "<u8"), ("sync_sample", "<u8"), ("eye", "<u4"),
                             ("a", "<u4"), ("b", "<u4"), ("inv_a", "<u4"), ("fiw_raw", "<u4"), ("fiw", "<u4"),
                             ("fiw_rc", "<u4"), ("sample_range", "<i4"), ("sample_delta", "<i4"), ("cycle", "<u4"),
                             ("frame", "<u4"), ("pad", "<u4"), ("words", "<u4", (4, 88))])
FLEX_MSG_DTYPE = np.dtype([("kind", "<u4"), ("baud", "<u4"), ("phase", "<u4"), ("cycle", "<u4"), ("frame", "<u4"),
                           ("aux0", "<u4"), ("aux1", "<u4"), ("aux2", "<u4"), ("capcode", "<u8"), ("sample", "<u8"),
                           ("len", "<u4"), ("pad", "<u4"), ("text", "S256")])
FLEX_EV_FRAME, FLEX_EV_BAD_BAUD, FLEX_EV_BAD_FIW = 1, 2, 3
FLEX_MSG_ALNUM, FLEX_MSG_NUM, FLEX_MSG_SIV = 1, 2, 3


def flex_msg_tuple(m, with_sample=True):
    """(kind, baud, phase, cycle, frame, aux0, aux1, aux2, capcode, text[, sample])"""
    n = int(m["len"])
    raw = bytes(m["text"]).ljust(256, b"\0")[:n]
    t = (int(m["kind"]), int(m["baud"]), int(m["phase"]), int(m["cycle"]), int(m["frame"]), int(m["aux0"]), int(m["aux1"]),
         int(m["aux2"]), int(m["capcode"]), raw)
    return t + (int(m["sample"]),) if with_sample else t


class FlexCoding(C.Structure):
    _fields_ = [("seq_a", C.c_uint16), ("baud", C.c_uint16), ("fsk_levels", C.c_uint8), ("sample_skip", C.c_uint8),
                ("sync_2_samples", C.c_uint8), ("sym_bits", C.c_uint8), ("sample_fudge", C.c_uint8), ("nr_phases", C.c_uint8),
                ("symbols_per_block", C.c_uint16)]


def flex_coding(idx):
    f = lib().mfmo_flex_coding
    f.argtypes = [C.c_uint]
    f.restype = C.POINTER(FlexCoding)
    p = f(idx)
    return p.contents if p else None


class Flex:
    """Oracle FLEX decoder for one channel: feed(pcm) -> (events, messages) of that call."""

    def __init__(self):
        self.h = lib().mfmo_flex_new()
        assert self.h

    def feed(self, pcm):
        x = np.ascontiguousarray(pcm, dtype=np.int16)
        cap_ev = x.size // 1000 + 8
        cap_ms = 64 * (x.size // 28000 + 2)
        ev = np.zeros(cap_ev, FLEX_EVENT_DTYPE)
        ms = np.zeros(cap_ms, FLEX_MSG_DTYPE)
        nev, nms = C.c_size_t(0), C.c_size_t(0)
        if x.size:
            lib().mfmo_flex_on_pcm(self.h, p16(x), x.size, ev.ctypes.data, cap_ev, C.byref(nev), ms.ctypes.data, cap_ms,
                                   C.byref(nms))
        assert nev.value <= cap_ev and nms.value <= cap_ms
        return ev[:nev.value].copy(), [flex_msg_tuple(m) for m in ms[:nms.value]]

    def close(self):
        if self.h:
            lib().mfmo_flex_free(self.h)
            self.h = None

    def __del__(self):
        self.close()


def flex_phase_process(words, coding, phase, cycle, frame, sample=0):
    """oracle message layer on the 88 words of one phase -> (words after the in-place corrections, messages)"""
    w = np.ascontiguousarray(words, dtype=np.uint32).copy()
    assert w.size == 88
    ms = np.zeros(128, FLEX_MSG_DTYPE)
    n = C.c_size_t(0)
    r = lib().mfmo_flex_phase_process(w.ctypes.data_as(C.POINTER(C.c_uint32)), coding, phase, cycle, frame, sample,
                                      ms.ctypes.data, 128, C.byref(n))
    assert r == 0 and n.value <= 128
    return w, [flex_msg_tuple(m) for m in ms[:n.value]]


def unpack_bytes(raw, fmt):
    """one read of 8-bit samples -> int16, as file_if.c (fmt 1 cs8, 2 cu8) / rtl_sdr_if.c (fmt 3) widen it"""
    a = np.ascontiguousarray(raw).view(np.uint8).reshape(-1)
    out = np.zeros(a.size, np.int16)
    lib().mfmo_unpack_bytes(a.ctypes.data, a.size, fmt, p16(out))
    return out


class F32Channel:
    """oracle/f32_oracle.c: the fp64 restatement of the floating-point IQ path, one channel."""

    def __init__(self, offset_hz, sample_rate, decimation, lpf_taps, gain=1.0):
        t = np.ascontiguousarray(lpf_taps, dtype=np.float64)
        self.nt, self.decim = t.size, decimation
        self.h = lib().mfmo_f32_chan_new(int(offset_hz), sample_rate, decimation,
                                         t.ctypes.data_as(C.POINTER(C.c_double)), t.size, float(gain))

    def taps(self):
        re, im = np.zeros(self.nt), np.zeros(self.nt)
        lib().mfmo_f32_chan_taps(self.h, re.ctypes.data_as(C.POINTER(C.c_double)), im.ctypes.data_as(C.POINTER(C.c_double)))
        return re, im

    def push(self, iq):
        """iq float32 [n][2] -> (pcm float64 [m], iq float64 [m][2])"""
        a = np.ascontiguousarray(iq, dtype=np.float32).reshape(-1)
        cap = a.size // 2 // self.decim + 4
        pcm, out = np.zeros(cap), np.zeros((cap, 2))
        n = lib().mfmo_f32_chan_push(self.h, a.ctypes.data_as(C.POINTER(C.c_float)), a.size // 2,
                                     pcm.ctypes.data_as(C.POINTER(C.c_double)), out.ctypes.data_as(C.POINTER(C.c_double)), cap)
        return pcm[:n].copy(), out[:n].copy()

    def close(self):
        if self.h:
            lib().mfmo_f32_chan_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MuellerMuller:
    """oracle/pocsag_oracle.c: mm_init / mm_process (pager/mueller_muller.c), one channel."""

    def __init__(self, kw, km, samples_per_bit, error_min, error_max):
        self.st = (C.c_float * 10)()
        lib().mfmo_mm_init(self.st, kw, km, samples_per_bit, error_min, error_max)

    def process(self, buf, offset, nr_samples):
        """decisions of samples[offset : offset + nr_samples] of `buf` (the sample behind the slice is read when the
        loop asks for it, as in the reference's test)"""
        a = np.ascontiguousarray(buf, dtype=np.int16)
        cap = nr_samples + 16
        dec = np.zeros(cap, np.int16)
        ptr = C.cast(a.ctypes.data + 2 * offset, C.POINTER(C.c_int16))
        n = lib().mfmo_mm_process(self.st, ptr, nr_samples, p16(dec), cap)
        return dec[:n].copy()


def window_pcm(iq, cre, cim, decimation, incr, first_sample_is_output, w0, count):
    """PCM of outputs [w0, w0 + count) of one channel, w0 >= 1, computed from the samples of that window alone.  `iq` holds
    stream samples starting at the first sample of output `first_sample_is_output` (a global output index); w0 counts from
    there.  The rotator is stepped to output first_sample_is_output + w0 - 1, that output is computed for its filtered
    sample only (its PCM needs the sample before it) and dropped."""
    T = np.asarray(cre).size
    ch = Channel(cre, cim, decimation, incr)
    ch.skip_outputs(first_sample_is_output + w0 - 1)
    lo = (w0 - 1) * decimation
    hi = (w0 - 1 + count) * decimation + T
    pcm, _ = ch.feed(iq[lo:hi])
    ch.close()
    assert pcm.size == count + 1, (pcm.size, count)
    return pcm[1:]
