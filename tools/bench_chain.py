#!/usr/bin/env python3
"""Device-resident chain timing: wideband IQ (HBM) -> channel engine -> 4/5 resampler -> POCSAG stage, 64 channels.
Everything is queued on the engine's stream; the only host traffic is the event list at the end.

    python tools/bench_chain.py [--block-log2 24] [--iters 30]

For DESIGN.md (what the stages behind the channel kernel cost next to it); not part of bench.py's contract line."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--block-log2", type=int, default=24)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--channels", type=int, default=64)
    args = ap.parse_args()
    from __graft_entry__ import load_package
    pkg = load_package()
    sy = pkg.synth
    # 1.2 MS/s, D = 24 -> 50 kS/s PCM -> 4/5 ... keep the pager's 38 400 Hz exact: fs = 1 152 000, D = 24 -> 48 kS/s -> 4/5
    fs, decim, C = 1152000, 24, args.channels
    taps = sy.design_lpf(128, 12500.0, fs)
    offs = sy.channel_offsets(C, fs)
    blk = 1 << args.block_log2
    iq = sy.synth_iq(blk, fs, offs[:: max(1, C // 8)][:8], seed=5)
    eng = pkg.Engine(fs, decim, blk, device=0, flags=pkg.binding.MFM_F_DEVICE_ONLY)
    for o in offs:
        eng.add_channel(int(o), taps, 1.0)
    eng.commit()
    rtaps = (sy.design_lpf(81, 0.45 / 5, 1.0) * 4 * 16384).astype(np.int16)
    rs = pkg.Resampler(C, rtaps, 4, 5, blk // decim + 8, device=0)
    pg = pkg.Pocsag(C, rs.max_out(), device=0)

    def step(fetch):
        assert eng.push(iq) == 0
        dptr, stride, nout, _ = eng.last_output_device()
        yptr, ystride, ny = rs.process_device(dptr, stride, nout, stream=eng.stream)
        pg.process_device(yptr, ystride, ny, stream=eng.stream)
        if fetch:
            return len(pg.fetch_events())
        return 0

    for _ in range(5):
        step(True)
    eng.sync()
    t0 = time.perf_counter()
    nev = 0
    for i in range(args.iters):
        nev = step(i == args.iters - 1)
    eng.sync()
    dt = (time.perf_counter() - t0) / args.iters
    print(json.dumps({"chain": "IQ(host push) -> engine -> resampler 4/5 -> pocsag", "channels": C, "block_samples": blk,
                      "ms_per_block": round(dt * 1e3, 4), "input_msamples_per_s": round(blk / dt / 1e6, 1),
                      "msamp_per_s_x_channels": round(blk * C / dt / 1e6, 1), "events_last_block": nev,
                      "kernel": eng.stats()["kernel_variant"]}))
    eng.close()
    rs.close()
    pg.close()


if __name__ == "__main__":
    main()
