#!/usr/bin/env python3
"""Regenerate the committed golden fixtures.  Run in the authoring container (needs /root/reference
for the reference-built part; `make -C oracle` first).

  atan2_ref.npz   inputs (y, x) and the bit patterns returned by the REFERENCE's own fast_atan2f
                  object code (oracle/_ref/libref_fast_atan2f.so, built from
                  /root/reference/multifm/fast_atan2f.c).  This is what pins the oracle's fast_atan2f.
  path_oracle.npz a small end-to-end vector (IQ in -> filtered IQ + PCM out, 3 channels) produced by
                  the ORACLE.  It is a regression pin for the restatement, not reference output: the
                  reference's FIR / discriminator sources need TSL headers the image lacks.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import oracle_lib as ora  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402


def atan2_inputs(seed=11):
    rng = np.random.RandomState(seed)
    parts = []
    # the discriminator feeds (float)int32 values: random ints over several magnitudes
    for bits in (4, 8, 12, 16, 20, 24, 28, 31):
        lim = 1 << bits
        a = rng.randint(-lim, lim, size=(1500, 2)).astype(np.float32)
        parts.append(a)
    # table knots and their neighbours: y/x = k/255 exactly and +-1 ulp
    k = np.arange(0, 256, dtype=np.float32)
    for sx, sy in ((1, 1), (-1, 1), (1, -1), (-1, -1)):
        parts.append(np.stack([sy * k, sx * np.full(256, 255, np.float32)], axis=1))
        parts.append(np.stack([sy * np.full(256, 255, np.float32), sx * k], axis=1))
    # small-angle threshold neighbourhood, zeros, equal magnitudes
    z = np.float32(0.003921569)
    near = np.array([np.nextafter(z, np.float32(0)), z, np.nextafter(z, np.float32(1))], np.float32)
    for v in near:
        for sx, sy in ((1, 1), (-1, 1), (1, -1), (-1, -1)):
            parts.append(np.array([[sy * v, sx * 1.0], [sy * 1.0, sx * v]], np.float32))
    parts.append(np.array([[0, 0], [0, 1], [1, 0], [0, -1], [-1, 0], [5, 5], [-5, 5], [5, -5], [-5, -5],
                           [-2147483648.0, -2147483648.0], [2147483648.0, 1.0]], np.float32))
    # generic floats
    parts.append((rng.standard_normal((3000, 2)) * 1e6).astype(np.float32))
    return np.concatenate(parts, axis=0)


def main():
    ref = ora.ref_atan2()
    if ref is None:
        raise SystemExit("oracle/_ref/libref_fast_atan2f.so missing (needs /root/reference; run make -C oracle)")
    yx = atan2_inputs()
    out = np.array([ref.fast_atan2f(float(y), float(x)) for y, x in yx], dtype=np.float32)
    np.savez_compressed(os.path.join(HERE, "atan2_ref.npz"), yx=yx, bits=out.view(np.uint32))
    print("atan2_ref.npz:", yx.shape[0], "vectors from the reference object code")

    pkg = load_package()
    fs, decim = 2400000, 96
    taps = pkg.synth.design_lpf(128, 12500.0, fs)
    offs = [101000, -320000, 37500]
    gains = [1.0, 2.5118864315095806, 1.0]
    cre = np.stack([ora.make_taps(taps, o, fs, g)[0] for o, g in zip(offs, gains)])
    cim = np.stack([ora.make_taps(taps, o, fs, g)[1] for o, g in zip(offs, gains)])
    incr = np.stack([ora.rot_incr(o, fs, decim) for o in offs])
    iq = pkg.synth.synth_iq(96 * 300 + 128, fs, offs, seed=5)
    iq[1000:1400] = pkg.synth.random_iq(400, seed=9)  # a full-scale burst: int32 wrap-around
    pcm, q = ora.run_channels(iq, cre, cim, incr, decim, want_iq=True)
    np.savez_compressed(os.path.join(HERE, "path_oracle.npz"), fs=fs, decim=decim, lpf=taps, offsets=np.array(offs),
                        gains=np.array(gains), cre=cre, cim=cim, incr=incr, iq=iq, pcm=pcm, filt_iq=q)
    print("path_oracle.npz:", pcm.shape, "PCM from the oracle (unpinned regression vector)")


if __name__ == "__main__":
    main()
