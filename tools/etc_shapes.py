#!/usr/bin/env python3
"""Kernel time of the channelizer geometries the reference ships under etc/ (sample rate, decimation, filter length as in
tests/test_gpu_parity.py::ETC_CONFIGS), 64 channels each, blocks resident in HBM: which kernel runs and what a (channel,
output) costs.   python tools/etc_shapes.py [--channels 64]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SHAPES = [("multifm + flex_25khz_lpf", 1000000, 40, 128, 12500.0, 1 << 24),
          ("multifm_airspy + flex_25khz_lpf_3mhz", 3000000, 120, 512, 12500.0, 1 << 24),
          ("pocsag_rtlsdr + pocsag_1200khz_fs", 1200000, 25, 256, 9000.0, 1 << 24),
          ("pocsag_rtlsdr, 128 taps", 1200000, 25, 128, 9000.0, 1 << 24),
          ("pocsag_airspy + pocsag_narrow", 2500000, 100, 256, 4800.0, 1 << 24),
          ("multifm_file (no decimation)", 8738133, 1, 128, 400000.0, 1 << 20),
          ("cfg2 (bench headline)", 2400000, 96, 128, 12500.0, 1 << 24)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--channels", type=int, default=64)
    args = ap.parse_args()
    from __graft_entry__ import load_package
    pkg = load_package()
    b = pkg.binding
    rng = np.random.RandomState(1)
    print("%-40s %9s %5s %5s %7s %3s %9s %12s" % ("shape", "fs", "D", "taps", "block", "k", "kernel_ms", "ps/(ch,out)"))
    for name, fs, decim, ntaps, cutoff, block in SHAPES:
        taps = pkg.synth.design_lpf(ntaps, cutoff, fs)
        offs = rng.randint(-fs // 2 + 20000, fs // 2 - 20000, size=args.channels)
        eng = pkg.Engine(fs, decim, block, device=0, flags=b.MFM_F_DEVICE_ONLY | b.MFM_F_TIMING)
        for o in offs:
            eng.add_channel(int(o), taps, 1.0)
        eng.commit()
        # DEVICE_ONLY: no PCM mirror; two pushes fill both input buffers, replay() then launches on what they hold
        iq = pkg.synth.random_iq(min(block, 1 << 20), seed=3, full_scale=False)
        host = np.tile(iq, (-(-block // iq.shape[0]), 1))[:block].reshape(-1)
        for _ in range(2):
            assert eng.push(host) == 0
            eng.sync()
        for _ in range(12):
            eng.replay(block, 1)
        eng.sync()
        ms = eng.launch_ms(last=10)
        st = eng.stats()
        k = float(np.median(ms))
        eng.close()
        print("%-40s %9d %5d %5d %7s %3d %9.4f %12.2f" % (name, fs, decim, ntaps, "2^%d" % int(np.log2(block)), st["kernel_variant"], k,
                                                       k * 1e9 / (args.channels * (block / decim))),
              {x: st[x] for x in ("k_steps", "tap_hi_mask", "taps_resident", "lds_bytes", "outputs_per_tile", "grid_last") if x in st})


if __name__ == "__main__":
    main()
