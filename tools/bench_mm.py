"""Timing of the Mueller-Muller stage (mfm_mm_process_device) on resident PCM with the constants of the reference's
test (pager/test/test_mueller_muller.c:85-88: 1200 baud at 25 kHz).  One JSON line.

    python tools/bench_mm.py [--channels 64] [--samples 699050] [--iters 10]

Not part of bench.py's contract line."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--channels", type=int, default=64)
    ap.add_argument("--samples", type=int, default=699050)
    ap.add_argument("--iters", type=int, default=10)
    args = ap.parse_args()
    import torch
    from __graft_entry__ import load_package
    pkg = load_package()
    sy = pkg.synth
    C, n = args.channels, args.samples
    spb = np.float32(25000.0) / np.float32(1200.0)
    msgs = [(0x12345, 3, 2, sy.pocsag_alpha_words("THE QUICK BROWN FOX JUMPS OVER THE LAZY DOG " * 3))] * 8
    burst = sy.pocsag_pcm(sy.pocsag_bits(sy.pocsag_batches(msgs)), 1200, noise=900, lead=500, trail=500, seed=1, rate=25000)
    host = np.stack([np.roll(np.resize(burst, n + 8), 131 * c) for c in range(C)])
    x = torch.from_numpy(host).to("cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    mm = pkg.MuellerMuller(C, 0.0001, 0.000004, float(spb), float(spb - np.float32(0.05)), float(spb + np.float32(0.05)),
                           n, device=0)
    for _ in range(2):
        _, _, d_cnt = mm.process_device(x.data_ptr(), n + 8, n, stream=stream)
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(args.iters):
        mm.process_device(x.data_ptr(), n + 8, n, stream=stream)
    t1.record()
    torch.cuda.synchronize()
    ms = t0.elapsed_time(t1) / args.iters
    print(json.dumps({"stage": "mueller_muller", "channels": C, "pcm_samples_per_channel": n, "ms_per_block": round(ms, 4),
                      "pcm_msamples_per_s": round(C * n / ms / 1e3, 1), "decisions_per_channel": round(n / float(spb))}),
          flush=True)
    mm.close()


if __name__ == "__main__":
    main()
