#!/usr/bin/env python3
"""Phase timeline of the MFMA kernel from the -DMFM_TRACE build (make -C tsl-sdr_amd trace).
Run on the GPU box:  MFM_LIB=tsl-sdr_amd/build_trace/libmultifm_hip_trace.so python tools/trace_phases.py"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MFM_LIB", os.path.join(ROOT, "tsl-sdr_amd", "build_trace", "libmultifm_hip_trace.so"))
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
fs, decim, taps, offs, gains = pkg.synth.plan("cfg2_64ch")
block = 1 << 24
eng = pkg.Engine(fs, decim, block, device=0, flags=pkg.binding.MFM_F_DEVICE_ONLY)
for o, g in zip(offs, gains):
    eng.add_channel(int(o), taps, float(g))
eng.commit()
for _ in range(4):
    eng.acquire_input()
    eng.submit(block, wait_producer=False)
eng.sync()
buf = np.zeros(64 * 128, np.uint64)
assert eng.lib.mfm_trace_read(buf.ctypes.data_as(C.POINTER(C.c_uint64))) == 0
buf = buf.reshape(64, 64, 2)
names = {1: "start", 2: "prologue done", 3: "barrier1", 4: "staged", 5: "barrier2", 6: "mfma done", 7: "stores issued", 8: "epilogue done", 9: "staged (lds)"}
for wg in (0, 1, 8, 63):
    t0 = int(buf[wg, 0, 1])
    print(f"--- workgroup {wg}")
    prev = t0
    for k in range(60):
        i, t = int(buf[wg, k, 0]), int(buf[wg, k, 1])
        if i == 0:
            break
        print(f"  {names.get(i, i):14s} +{t - prev:7d}  (t={t - t0})")
        prev = t
