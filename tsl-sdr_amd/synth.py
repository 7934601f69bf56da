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
