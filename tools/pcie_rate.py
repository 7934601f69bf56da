#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer path (mfm_engine_push / mfm_engine_push_bytes + fetch): what a front
end delivering sample_bufs from host memory gets, as opposed to bench.py's HBM-resident figure.  With --u8 the
same blocks are also fed as raw 8-bit pairs (RTL-SDR format) and widened on the device."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
pkg = load_package()
fs, decim, taps, offs, gains = pkg.synth.plan("cfg2_64ch")
formats = [("cs16", 0, 4)] + ([("u8 (rtl-sdr)", 3, 2)] if "--u8" in sys.argv else [])
for blk_log2 in (17, 20, 24):
    for name, fmt, bytes_per_sample in formats:
        blk = 1 << blk_log2
        eng = pkg.Engine(fs, decim, blk, device=0)
        for o, g in zip(offs, gains):
            eng.add_channel(int(o), taps, float(g))
        eng.commit()
        iq = pkg.synth.random_iq(blk, seed=1, full_scale=False)
        raw8 = np.random.RandomState(2).randint(0, 256, size=(blk, 2)).astype(np.uint8)
        nblk = max(8, (1 << 27) // blk)
        def run():
            for _ in range(nblk):
                while (eng.push(iq) if fmt == 0 else eng.push_bytes(raw8, fmt)) == pkg.binding.MFM_E_BUSY:
                    while eng.fetch() is not None:
                        pass
            while eng.fetch() is not None:
                pass
        run()
        t0 = time.perf_counter(); run(); dt = time.perf_counter() - t0
        print(f"block 2^{blk_log2} {name:13s}: {nblk * blk / dt / 1e6:9.1f} MSamp/s input incl. H2D + D2H of PCM "
              f"({nblk * blk * bytes_per_sample / dt / 1e9:.2f} GB/s over PCIe), x64 channels = {nblk * blk * 64 / dt / 1e6:.0f} MSamp/s x ch",
              flush=True)
        eng.close()
