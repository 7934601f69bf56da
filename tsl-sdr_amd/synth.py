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
    "pocsag_rtlsdr": dict(fs=1200000, decim=25, taps=128, cutoff=12500.0, offsets=[-320000, -492000],
                          gains_db=[4.0, 0.0]),
    # the FLEX 25 kHz LPF of configs[4] (512 taps, etc/flex_25khz_lpf*.json) at the 2.4 MS/s / D = 96 geometry
    "cfg2_64ch_512taps": dict(fs=2400000, decim=96, taps=512, cutoff=12500.0, nr_channels=64),
    "cfg2_64ch_256taps": dict(fs=2400000, decim=96, taps=256, cutoff=12500.0, nr_channels=64),
    "cfg5_airspy": dict(fs=10000000, decim=400, taps=512, cutoff=12500.0, nr_channels=2048),
}


def plan(name, nr_channels=None):
    """Resolve a named configuration to (fs, decimation, lpf_taps, offsets[int32], gains[float])."""
    c = dict(CONFIGS[name])
    fs, decim = c["fs"], c["decim"]
    taps = design_lpf(c["taps"], c["cutoff"], fs)
    if "offsets" in c and nr_channels is None:
        offs = np.array(c["offsets"], dtype=np.int32)
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
