"""Synthetic wideband IQ and channel plans for the BASELINE.json configurations.

Everything here is deterministic (seeded) so tests, bench.py and the oracle see identical input.
Samples are interleaved int16 I,Q, exactly what a struct sample_buf carries
(reference filter/sample_buf.h:59-102).
"""
import numpy as np


def design_lpf(nr_taps=128, cutoff_hz=12500.0, sample_rate_hz=2400000.0):
    """Real low-pass taps, Hamming-windowed sinc, unity DC gain (the reference's tap files
    etc/flex_25khz_lpf*.json are unity-sum low-pass designs of 128 / 512 taps)."""
    n = np.arange(nr_taps, dtype=np.float64) - (nr_taps - 1) / 2.0
    fc = cutoff_hz / sample_rate_hz
    h = 2.0 * fc * np.sinc(2.0 * fc * n) * np.hamming(nr_taps)
    return h / h.sum()


def channel_offsets(nr_channels, sample_rate_hz=2400000, spacing_hz=None, unaligned=True):
    """Channel centre offsets from the tuner frequency.

    nr_channels <= 64: a 37.5 kHz grid starting at -1.18125 MHz; a few
    slots are moved off-grid so the recursive rotator takes its decaying, non-periodic-looking path
    (offset*D/fs not a multiple of 1/4).  Larger sets pack the same span more densely (config 3).
    """
    span = 0.984375 * sample_rate_hz
    if spacing_hz is None:
        spacing_hz = 37500.0 if nr_channels <= 64 else span / nr_channels
    start = -span / 2.0
    offs = np.array([int(round(start + spacing_hz * k)) for k in range(nr_channels)], dtype=np.int64)
    if unaligned and nr_channels >= 4:
        moves = {1: 101000, 2: 3125, nr_channels // 2: 777, nr_channels - 1: -433219}
        for k, v in moves.items():
            offs[k] = v
    return offs.astype(np.int32)


def synth_iq(nr_samples, sample_rate_hz, carrier_offsets_hz, seed=7, amplitude=6000.0, noise=512,
             deviation_hz=4500.0, tone_hz=1200.0):
    """Sum of FM carriers (deviation 4.5 kHz, a tone per carrier) plus uniform noise, clipped to int16.
    Returns an int16 array of shape (nr_samples, 2)."""
    rng = np.random.RandomState(seed)
    carriers = np.asarray(carrier_offsets_hz, dtype=np.float64).reshape(-1)
    out_i = np.zeros(nr_samples, dtype=np.float32)
    out_q = np.zeros(nr_samples, dtype=np.float32)
    amp = amplitude / max(1.0, np.sqrt(len(carriers)))
    chunk = 1 << 20
    for ci, f0 in enumerate(carriers):
        tone = tone_hz * (1.0 + 0.07 * ci)
        beta = deviation_hz / tone
        for s in range(0, nr_samples, chunk):
            t = (np.arange(s, min(s + chunk, nr_samples), dtype=np.float64)) / sample_rate_hz
            ph = 2.0 * np.pi * f0 * t + beta * np.sin(2.0 * np.pi * tone * t) + 0.37 * ci
            out_i[s:s + len(t)] += (amp * np.cos(ph)).astype(np.float32)
            out_q[s:s + len(t)] += (amp * np.sin(ph)).astype(np.float32)
    iq = np.empty((nr_samples, 2), dtype=np.int16)
    ni = rng.randint(-noise, noise + 1, size=nr_samples)
    nq = rng.randint(-noise, noise + 1, size=nr_samples)
    iq[:, 0] = np.clip(np.rint(out_i) + ni, -32768, 32767).astype(np.int16)
    iq[:, 1] = np.clip(np.rint(out_q) + nq, -32768, 32767).astype(np.int16)
    return iq


def random_iq(nr_samples, seed=1, full_scale=True):
    """Uniform random int16 IQ (stresses int32 wrap-around and every atan2 octant)."""
    rng = np.random.RandomState(seed)
    lim = 32768 if full_scale else 4096
    return rng.randint(-lim, lim, size=(nr_samples, 2)).astype(np.int16)


CONFIGS = {
    # BASELINE.json configs[0..4] restated concretely
    "multifm_1ch": dict(fs=1000000, decim=40, taps=128, cutoff=12500.0, offsets=[112500]),
    "multifm_1ch_2400k": dict(fs=2400000, decim=96, taps=128, cutoff=12500.0, offsets=[112500]),
    "cfg2_64ch": dict(fs=2400000, decim=96, taps=128, cutoff=12500.0, nr_channels=64),
    "cfg3_1024ch": dict(fs=2400000, decim=96, taps=128, cutoff=12500.0, nr_channels=1024),
    # the same geometry with every channel on the 12.5 kHz raster: all rotators exact
    "cfg2_64ch_grid": dict(fs=2400000, decim=96, taps=128, cutoff=12500.0, nr_channels=64, grid=True),
    "cfg3_1024ch_grid": dict(fs=2400000, decim=96, taps=128, cutoff=12500.0, nr_channels=1024, grid=True),
    "pocsag_rtlsdr": dict(fs=1200000, decim=25, taps=128, cutoff=12500.0, offsets=[-320000, -492000],
                          gains_db=[4.0, 0.0]),
    # the same channelizer with the low-pass the reference ships for that sample rate (etc/pocsag_1200khz_fs.json: 256 taps,
    # hamming, 9 kHz), etc/pocsag_airspy.json + etc/pocsag_narrow.json (2.5 MS/s, D = 100, 256 taps, 4.8 kHz), and
    # etc/multifm_airspy.json / multifm_usrp.json + etc/flex_25khz_lpf_3mhz.json (3 MS/s, D = 120, 512 taps)
    "pocsag_rtlsdr_256taps": dict(fs=1200000, decim=25, taps=256, cutoff=9000.0, offsets=[-320000, -492000], gains_db=[4.0, 0.0]),
    "pocsag_airspy": dict(fs=2500000, decim=100, taps=256, cutoff=4800.0, offsets=[-320000], gains_db=[2.0]),
    "multifm_airspy": dict(fs=3000000, decim=120, taps=512, cutoff=12500.0, offsets=[-887500]),
    # the FLEX 25 kHz LPF of configs[4] (512 taps, etc/flex_25khz_lpf*.json) at the 2.4 MS/s / D = 96 geometry
    "cfg2_64ch_512taps": dict(fs=2400000, decim=96, taps=512, cutoff=12500.0, nr_channels=64),
    "cfg2_64ch_256taps": dict(fs=2400000, decim=96, taps=256, cutoff=12500.0, nr_channels=64),
    "cfg5_airspy": dict(fs=10000000, decim=400, taps=512, cutoff=12500.0, nr_channels=2048),
}


def grid_offsets(nr_channels, sample_rate_hz=2400000, decimation=96):
    """Channel centres on the grid of half the output rate (12.5 kHz at 2.4 MS/s / 96), wrapping around inside the band:
    every rotator increment is exactly (16384, 0) or (-16384, 0) (filter/direct_fir.c:72-79), the channel plans real
    deployments use (a 25 kHz or 12.5 kHz raster)."""
    half = sample_rate_hz // decimation // 2
    span = int(0.99 * sample_rate_hz / 2) // half
    return np.array([half * ((k % (2 * span)) - span) for k in range(nr_channels)], dtype=np.int32)


def plan(name, nr_channels=None):
    """Resolve a named configuration to (fs, decimation, lpf_taps, offsets[int32], gains[float])."""
    c = dict(CONFIGS[name])
    fs, decim = c["fs"], c["decim"]
    taps = design_lpf(c["taps"], c["cutoff"], fs)
    if "offsets" in c and nr_channels is None:
        offs = np.array(c["offsets"], dtype=np.int32)
    elif c.get("grid"):
        offs = grid_offsets(nr_channels or c["nr_channels"], fs, decim)
    else:
        offs = channel_offsets(nr_channels or c["nr_channels"], fs)
    gains_db = c.get("gains_db", [0.0] * len(offs))
    gains = np.array([10.0 ** (g / 10.0) for g in gains_db] + [1.0] * (len(offs) - len(gains_db)))
    return fs, decim, taps, offs, gains


# ---- synthetic POCSAG ----------------------------------------------------------
# Words are built in the bit order the reference holds them in (pager/pager_pocsag.c:477 fills batch words LSB
# first): bit 0 = address/message flag, bits 1..20 = payload, bits 21..30 = BCH(31,21) parity (bit 30 - j is the
# coefficient of x^j, pager/bch_code.c:325), bit 31 = even parity.  On the air that is a standard POCSAG codeword,
# MSB first.

POCSAG_SYNC = 0x7CD215D8   # pager/pager_pocsag_priv.h:40, sent MSB first
POCSAG_IDLE = 0x6983915E   # pager/pager_pocsag_priv.h:46 (31 bits, as held after the mask)
_BCH_G = 0x769             # x^10 + x^9 + x^8 + x^6 + x^5 + x^3 + 1


def pocsag_codeword(payload21):
    """payload21: bits 0..20 of the word -> full 32-bit word with BCH parity and the even-parity bit"""
    w = int(payload21) & 0x1FFFFF
    # data polynomial: bit b is the coefficient of x^(30 - b); reduce modulo g(x)
    rem = 0
    for b in range(21):
        rem = (rem << 1) | ((w >> b) & 1)          # highest power first
        if rem & (1 << 10):
            rem ^= _BCH_G
    for _ in range(10):                            # multiply by x^10
        rem <<= 1
        if rem & (1 << 10):
            rem ^= _BCH_G
    for j in range(10):                            # remainder coefficient of x^j -> bit 30 - j
        if (rem >> j) & 1:
            w |= 1 << (30 - j)
    if bin(w).count("1") & 1:
        w |= 1 << 31
    return w


def pocsag_address_word(addr18, function):
    return pocsag_codeword(((int(addr18) & 0x3FFFF) << 1) | ((int(function) & 3) << 19))


def pocsag_data_words(bits, pad=(0,)):
    """bits: list of 0/1 payload bits in transmit order -> message codewords (20 bits each, padded with the
    repeating `pad` pattern)"""
    words = []
    bits = list(bits)
    while len(bits) % 20:
        bits.append(pad[(len(bits)) % len(pad)])
    for i in range(0, len(bits), 20):
        chunk = bits[i:i + 20]
        val = sum(int(chunk[k]) << k for k in range(20))
        words.append(pocsag_codeword(1 | (val << 1)))
    return words


def pocsag_alpha_words(text):
    bits = []
    for ch in text.encode("ascii"):
        bits += [(ch >> k) & 1 for k in range(7)]  # 7-bit characters, LSB first (pager_pocsag.c:378-399)
    return pocsag_data_words(bits)


def pocsag_numeric_words(digits):
    table = "0123456789XU -[]"                      # pager_pocsag.c:299-316
    bits = []
    for ch in digits:
        v = table.index(ch)
        bits += [(v >> k) & 1 for k in range(4)]
    return pocsag_data_words(bits, pad=(0, 0, 1, 1))   # fill with the "space" digit 0xC


def pocsag_batches(messages):
    """messages: list of (addr18, frame, function, data_words).  Each address word goes into its frame's first
    slot (word 2 * frame) of the next batch that still has it free; the data words follow; the rest is idle."""
    idle = POCSAG_IDLE | ((bin(POCSAG_IDLE).count("1") & 1) << 31)
    stream = []                                     # flat list of words, 16 per batch
    for addr18, frame, function, data in messages:
        while len(stream) % 16 != 2 * frame:
            stream.append(idle)
        stream.append(pocsag_address_word(addr18, function))
        stream += list(data)
    stream.append(idle)
    while len(stream) % 16:
        stream.append(idle)
    return [stream[i:i + 16] for i in range(0, len(stream), 16)]


def pocsag_bits(batches, preamble_bits=576):
    bits = [(i + 1) & 1 for i in range(preamble_bits)]          # 1010...
    for batch in batches:
        bits += [(POCSAG_SYNC >> (31 - k)) & 1 for k in range(32)]
        for w in batch:
            bits += [(int(w) >> k) & 1 for k in range(32)]
    return np.array(bits, dtype=np.uint8)


def pocsag_pcm(bits, baud, amplitude=8000, noise=0.0, lead=0, trail=0, seed=0, flip=None, rate=38400):
    """NRZ PCM at `rate` Hz (38 400 is what the decoder wants): a 1 is a negative sample (pager_pocsag.c:91).
    `flip`: indices of bits to invert (channel errors).  lead / trail: noise-only samples before / after."""
    b = np.array(bits, dtype=np.int64)
    if flip is not None and len(flip):
        b[np.asarray(flip)] ^= 1
    rng = np.random.RandomState(seed)
    n = (b.size * rate + baud - 1) // baud
    idx = np.minimum((np.arange(n, dtype=np.int64) * baud) // rate, b.size - 1)
    sig = np.where(b[idx] == 1, -amplitude, amplitude).astype(np.float64)
    x = np.concatenate([np.zeros(lead), sig, np.zeros(trail)])
    if noise > 0:
        x = x + rng.normal(0.0, noise, size=x.size)
    return np.clip(np.round(x), -32768, 32767).astype(np.int16)


def pocsag_fm_iq(bits, baud, sample_rate_hz, carrier_hz, deviation_hz=4500.0, amplitude=9000.0, lead=0, trail=0,
                 noise=200.0, seed=0):
    """2-FSK at `carrier_hz` from the tuner centre: bit 1 -> -deviation (negative discriminator output)."""
    spb = int(round(sample_rate_hz / baud))
    f = np.concatenate([np.zeros(lead), np.repeat(np.where(np.asarray(bits) == 1, -deviation_hz, deviation_hz), spb),
                        np.zeros(trail)]) + carrier_hz
    ph = 2.0 * np.pi * np.cumsum(f) / sample_rate_hz
    rng = np.random.RandomState(seed)
    i = amplitude * np.cos(ph) + rng.normal(0.0, noise, size=ph.size)
    q = amplitude * np.sin(ph) + rng.normal(0.0, noise, size=ph.size)
    out = np.empty((ph.size, 2), np.int16)
    out[:, 0] = np.clip(np.round(i), -32768, 32767)
    out[:, 1] = np.clip(np.round(q), -32768, 32767)
    return out


# ---- synthetic FLEX ------------------------------------------------------------
# Written from the published frame layout, not from the decoder: 115.2 ms sync 1 at 1600 bit/s 2-FSK (32 bits of
# 1010.., A = mode code + 0x5939, B = 0x5555, inverted A, the frame information word), 25 ms sync 2 at the frame's
# own symbol rate, then 11 blocks of 160 ms; every block carries 8 words per phase, bit-interleaved, words LSB
# first.  Words use the same BCH(31,21) + parity layout as POCSAG (info = bits 0..20).  Numbers that have to agree
# with the reference are cited (pager/pager_flex.c).

FLEX_RATE = 16000                                   # pager_flex_priv.h:231: "The input for this must always be a 16kHz signal"
FLEX_CODINGS = (                                    # pager_flex.c:46-96
    dict(seq_a=0x78F3, baud=1600, levels=2, phases=(0,), sym_rate=1600, sync2_dots=4),
    dict(seq_a=0x84E7, baud=3200, levels=2, phases=(0, 2), sym_rate=3200, sync2_dots=24),
    dict(seq_a=0x4F97, baud=3200, levels=4, phases=(0, 2), sym_rate=1600, sync2_dots=12),
    dict(seq_a=0x215F, baud=6400, levels=4, phases=(0, 1, 2, 3), sym_rate=3200, sync2_dots=32),
)
FLEX_NUM_TABLE = "0123456789XU -]["                  # pager_flex.c:686-704

flex_codeword = pocsag_codeword


def flex_checksummed(info21):
    """set bits 0..3 so that the six nibbles of the 21-bit word add up to 15 modulo 16 (pager_flex.c:107-119)"""
    w = int(info21) & 0x1FFFF0
    s = sum((w >> (4 * n)) & 0xF for n in range(6))
    return w | ((0xF - s) & 0xF)


def flex_fiw(cycle, frame, roaming=0, repeat=0, traffic=0):
    return flex_codeword(flex_checksummed((cycle & 0xF) << 4 | (frame & 0x7F) << 8 | (roaming & 1) << 15 | (repeat & 1) << 16
                                          | (traffic & 0xF) << 17))


def flex_biw(vsw, eob=0, priority=0, carry=0, collapse=0):
    return flex_codeword(flex_checksummed((priority & 0xF) << 4 | (eob & 3) << 8 | (vsw & 0x3F) << 10 | (carry & 3) << 16
                                          | (collapse & 7) << 18))


def flex_extra_biw(function, payload14):
    return flex_codeword(flex_checksummed((function & 7) << 4 | (payload14 & 0x3FFF) << 7))


def _flex_pack(bits):
    """bit list -> 21-bit info values, LSB first"""
    out = []
    for i in range(0, len(bits), 21):
        chunk = bits[i:i + 21]
        out.append(sum(int(b) << k for k, b in enumerate(chunk)))
    return out


def flex_alnum_body(text, seq=3, fragment=False, maildrop=False, signature=0x2A):
    """message words of an alphanumeric page: header, then 3 characters per word; the first character slot of an
    initial fragment (seq 3) carries the signature.  ETX (0x03) fills the last word."""
    head = (int(fragment) << 10) | ((seq & 3) << 11) | (int(maildrop) << 20)
    chars = list(text.encode("ascii"))
    if seq == 3:
        chars = [signature & 0x7F] + chars
    while len(chars) % 3:
        chars.append(0x03)
    words = [head]
    for i in range(0, len(chars), 3):
        words.append(chars[i] | chars[i + 1] << 7 | chars[i + 2] << 14)
    return [flex_codeword(w) for w in words]


def flex_numeric_body(digits, k=0):
    """message words of a numeric page: 2 check bits, then 4-bit digits, packed 21 bits per word, filled with spaces"""
    bits = [(k >> i) & 1 for i in range(2)]
    for ch in digits:
        v = FLEX_NUM_TABLE.index(ch)
        bits += [(v >> i) & 1 for i in range(4)]
    total = 21 * ((len(bits) + 20) // 21)
    while len(bits) < total:
        bits += [0, 0, 1, 1]                         # the "space" digit 0xC, LSB first
    return [flex_codeword(w) for w in _flex_pack(bits[:total])]


_FLEX_FILL = tuple(int(v) for v in np.random.RandomState(88).randint(0, 1 << 21, 88))


def flex_phase_words(records, eob=0, extra_biws=(), idle=_FLEX_FILL):
    """records: list of dicts
         kind   'alnum' | 'numeric' | 'tone' | 'siv' | 'raw'
         capcode (short address) or long=(first21, second21)
         alnum: text, seq, fragment, maildrop;  numeric: digits;  tone: digits (3, or 8 with a long address),
         ttype; siv: siv_type, data;  raw: vtype (vector type), body (list of info words)
       -> the 88 words of one phase: BIW, addresses, vectors, message words, fill.
       `idle`: information values the unused words cycle through.  (0, 0x1FFFFF) makes the interleaved block an
       alternating bit pattern - on a 1600 bit/s frame that is one long bit-sync-1 look-alike.)"""
    addr_start = 1 + eob
    addr_words, recs = [], []
    for r in records:
        if "long" in r:
            a = [flex_codeword(r["long"][0]), flex_codeword(r["long"][1])]
        else:
            a = [flex_codeword(int(r["capcode"]) + 32768)]
        recs.append((r, len(addr_words), len(a)))
        addr_words += a
    vsw = addr_start + len(addr_words)
    body_at = vsw + len(addr_words)
    words = {0: flex_biw(vsw, eob=eob)}
    for i, w in enumerate(extra_biws):
        words[1 + i] = w
    for i, w in enumerate(addr_words):
        words[addr_start + i] = w
    for r, aoff, alen in recs:
        kind = r["kind"]
        is_long = alen == 2
        if kind == "alnum":
            body = flex_alnum_body(r["text"], r.get("seq", 3), r.get("fragment", False), r.get("maildrop", False))
            vec = flex_checksummed(5 << 4 | (body_at & 0x7F) << 7 | (len(body) & 0x7F) << 14)
        elif kind == "numeric":
            body = flex_numeric_body(r["digits"])
            vec = flex_checksummed(3 << 4 | (body_at & 0x7F) << 7 | ((len(body) - 1) & 7) << 14)
        elif kind == "tone":
            d = [FLEX_NUM_TABLE.index(c) for c in r.get("digits", "000")]
            vec = flex_checksummed(2 << 4 | (r.get("ttype", 0) & 3) << 7 | d[0] << 9 | d[1] << 13 | d[2] << 17)
            body = []
            if is_long:
                rest = (d[3:] + [12] * 5)[:5]
                body = [flex_codeword(sum(v << (4 * i) for i, v in enumerate(rest)))]
        elif kind == "siv":
            vec = flex_checksummed(1 << 4 | (r["siv_type"] & 7) << 7 | (r["data"] & 0x7FF) << 10)
            body = [flex_codeword(0)] if is_long else []
        else:
            body = [flex_codeword(w) for w in r.get("body", [])]
            vec = flex_checksummed((r["vtype"] & 7) << 4 | (body_at & 0x7F) << 7 | (len(body) & 0x7F) << 14)
        words[vsw + aoff] = flex_codeword(vec)
        if is_long:
            # the second vector slot of a long address carries the first message word
            if not body:
                body = [flex_codeword(0)]
            words[vsw + aoff + 1] = body[0]
            body = body[1:]
        for w in body:
            words[body_at] = w
            body_at += 1
    if body_at > 88:
        raise ValueError("phase overflows 88 words")
    fill = [flex_codeword(v) for v in idle]
    return np.array([words.get(i, fill[i % len(fill)]) for i in range(88)], dtype=np.uint32)


def flex_interleave(words88):
    """88 words -> 2816 bits in transmit order: per block of 8 words, bit j of word 0..7, j = 0..31"""
    w = np.asarray(words88, dtype=np.uint64).reshape(11, 8)
    j = np.arange(32, dtype=np.uint64)
    return ((w[:, None, :] >> j[None, :, None]) & 1).astype(np.uint8).reshape(-1)   # [block][bit][word]


def flex_symbols(coding, phase_bits):
    """phase_bits: dict phase index -> 2816 bits.  Returns the block's symbols (0..3 for 4-FSK: bit 1 = phase A/C,
    bit 0 = phase B/D, as the slicer numbers them, pager_flex.c:148-171; 0/1 for 2-FSK)."""
    c = FLEX_CODINGS[coding]
    ph = [np.asarray(phase_bits[p], dtype=np.uint8) for p in c["phases"]]
    if len(ph) == 1:
        return ph[0]
    if len(ph) == 2 and c["levels"] == 2:
        return np.stack([ph[0], ph[1]], axis=1).reshape(-1)
    if len(ph) == 2:
        return ph[0] * 2 + ph[1]
    return np.stack([ph[0] * 2 + ph[1], ph[2] * 2 + ph[3]], axis=1).reshape(-1)


def flex_frame_levels(coding, cycle, frame, phases, corrupt=None, fiw_flip=0, a_flip=0):
    """One frame as a list of (level, samples) runs at 16 kHz; level in {-3, -1, +1, +3} thirds of the deviation.
    phases: dict phase index -> 88 words (missing phases are idle).  corrupt: {(phase, word): xor mask}."""
    c = FLEX_CODINGS[coding]
    a = ((c["seq_a"] << 16) | 0x5939) ^ a_flip
    bits = [(i + 1) & 1 for i in range(32)]                                   # 1010...10
    bits += [(a >> (31 - k)) & 1 for k in range(32)]
    bits += [(0x5555 >> (15 - k)) & 1 for k in range(16)]
    bits += [((~a) >> (31 - k)) & 1 for k in range(32)]
    fiw = flex_fiw(cycle, frame) ^ fiw_flip
    bits += [(fiw >> k) & 1 for k in range(32)]
    runs = [(3 if b else -3, FLEX_RATE // 1600) for b in bits]
    sps = FLEX_RATE // c["sym_rate"]
    two = c["levels"] == 2
    lv = (lambda s: 3 if s else -3) if two else (lambda s: (-3, -1, 3, 1)[s])
    # sync 2: dots, C, inverted dots, inverted C - the decoder only counts them (pager_flex.c:460-525)
    nsym_c = 16 if two else 8
    cpat = [(0xED84 >> (15 - k)) & 1 for k in range(16)]
    csym = cpat if two else [cpat[2 * k] * 2 + cpat[2 * k + 1] for k in range(8)]
    dots = [(k + 1) & 1 for k in range(c["sync2_dots"])]
    top = 1 if two else 2
    s2 = [top * d for d in dots] + csym[:nsym_c] + [top * (1 - d) for d in dots] + [(1 if two else 3) - s for s in csym[:nsym_c]]
    runs += [(lv(s), sps) for s in s2]
    pb = {}
    for p in c["phases"]:
        w = np.array(phases.get(p, flex_phase_words([])), dtype=np.uint32).copy()
        for (pp, wi), mask in (corrupt or {}).items():
            if pp == p:
                w[wi] ^= np.uint32(mask)
        pb[p] = flex_interleave(w)
    runs += [(lv(int(s)), sps) for s in flex_symbols(coding, pb)]
    return runs


def flex_levels(frames, lead=0, trail=0, gap=0):
    """frames -> one level per 16 kHz sample (0 outside the frames)"""
    parts = [np.zeros(lead)]
    for k, runs in enumerate(frames):
        parts.append(np.repeat(np.array([r[0] for r in runs], dtype=np.float64), [r[1] for r in runs]))
        if gap and k + 1 < len(frames):
            parts.append(np.zeros(gap))
    parts.append(np.zeros(trail))
    return np.concatenate(parts)


def flex_pcm(frames, amplitude=9000, noise=0.0, lead=0, trail=0, seed=0, offset=0, gap=0, rate=FLEX_RATE):
    """frames: list of flex_frame_levels() results, sent back to back (`gap` idle samples between them).
    A level of +-3 is +-amplitude (sync 1 always uses the outer levels).  `offset` adds a DC error.  `rate`: output
    sample rate (16 000 is what the decoder wants; 25 000 is what a multifm channel delivers)."""
    lv = flex_levels(frames, lead, trail, gap)
    if rate != FLEX_RATE:
        n = lv.size * rate // FLEX_RATE
        lv = lv[np.minimum((np.arange(n, dtype=np.int64) * FLEX_RATE) // rate, lv.size - 1)]
    x = lv * (amplitude / 3.0) + offset
    if noise > 0:
        x = x + np.random.RandomState(seed).normal(0.0, noise, size=x.size)
    return np.clip(np.round(x), -32768, 32767).astype(np.int16)


def flex_fm_iq(frames, sample_rate_hz, carrier_hz, deviation_hz=4800.0, amplitude=9000.0, lead=0, trail=0, noise=200.0, seed=0):
    """2- / 4-level FSK at `carrier_hz` from the tuner centre: level +-3 -> +-deviation, +-1 -> +-deviation / 3
    (a positive shift gives a positive discriminator output, i.e. a one).  lead / trail in 16 kHz samples."""
    lv = flex_levels(frames, lead, trail)
    n = int(lv.size * sample_rate_hz // FLEX_RATE)
    idx = np.minimum((np.arange(n, dtype=np.int64) * FLEX_RATE) // int(sample_rate_hz), lv.size - 1)
    f = lv[idx] * (deviation_hz / 3.0) + carrier_hz
    ph = 2.0 * np.pi * np.cumsum(f) / sample_rate_hz
    rng = np.random.RandomState(seed)
    out = np.empty((n, 2), np.float64)
    out[:, 0] = amplitude * np.cos(ph) + rng.normal(0.0, noise, size=n)
    out[:, 1] = amplitude * np.sin(ph) + rng.normal(0.0, noise, size=n)
    return out
