#!/usr/bin/env python3
"""Kernel time per (channel, output) of the second-generation kernel against the decimation (same 128-tap low-pass, 64
channels, 2^26-sample blocks resident in HBM): what part of a tile's time scales with the samples staged per output?"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
b = pkg.binding
import torch  # noqa: E402

block = 1 << 26
out = []
for decim in (25, 32, 40, 64, 96, 128):
    fs = 25000 * decim
    taps = pkg.synth.design_lpf(128, 12500.0, fs)
    offs = [int((k - 32) * 37500 * decim / 96) for k in range(64)]
    eng = pkg.Engine(fs, decim, block, device=0, flags=b.MFM_F_DEVICE_ONLY | b.MFM_F_TIMING)
    for o in offs:
        eng.add_channel(o, taps, 1.0)
    eng.commit()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.4:
        eng.replay(block, 8)
        eng.sync()
    eng.replay(block, 40)
    eng.sync()
    ms = float(np.mean(eng.launch_ms(40)))
    st = eng.stats()
    nout = block // decim
    eng.close()
    row = {"decimation": decim, "kernel_variant": st["kernel_variant"], "k_steps": st["k_steps"], "tap_hi_mask": st["tap_hi_mask"],
           "kernel_ms": ms, "ps_per_channel_output": ms * 1e9 / (64 * nout), "rot_exact_channels": st["rot_exact_channels"]}
    out.append(row)
    print(json.dumps(row), flush=True)
