#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer path (mfm_engine_push + fetch): what a front end delivering
sample_bufs from host memory gets, as opposed to bench.py's HBM-resident figure."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
pkg = load_package()
fs, decim, taps, offs, gains = pkg.synth.plan("cfg2_64ch")
for blk_log2 in (17, 20, 24):
    blk = 1 << blk_log2
    eng = pkg.Engine(fs, decim, blk, device=0)
    for o, g in zip(offs, gains):
        eng.add_channel(int(o), taps, float(g))
    eng.commit()
    iq = pkg.synth.random_iq(blk, seed=1, full_scale=False)
    nblk = max(8, (1 << 27) // blk)
    def run():
        done = 0
        for _ in range(nblk):
            while eng.push(iq) == pkg.binding.MFM_E_BUSY:
                while eng.fetch() is not None:
                    pass
        while eng.fetch() is not None:
            pass
    run()
    t0 = time.perf_counter(); run(); dt = time.perf_counter() - t0
    print(f"block 2^{blk_log2}: {nblk * blk / dt / 1e6:9.1f} MSamp/s input incl. H2D + D2H of PCM "
          f"({nblk * blk * 4 / dt / 1e9:.2f} GB/s over PCIe), x64 channels = {nblk * blk * 64 / dt / 1e6:.0f} MSamp/s x ch")
    eng.close()
