"""Timing of the pager stage (mfm_pocsag_process_device / mfm_flex_process_device) on resident PCM: idle channels
(sync search only), busy channels (back-to-back POCSAG batches / FLEX frames) and a mix.  One JSON line per scenario.

    python tools/bench_pager.py [--proto pocsag|flex] [--channels 64] [--samples 699050] [--iters 20]

(FLEX: 699 050 samples at 25 kS/s are 447 392 at 16 kHz.)

Used for DESIGN.md section 9 and profiles/r01_pager_*; not part of bench.py's contract line."""
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
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--proto", default="pocsag", choices=["pocsag", "flex"])
    args = ap.parse_args()
    import torch
    from __graft_entry__ import load_package
    pkg = load_package()
    sy = pkg.synth
    C, n = args.channels, args.samples
    rng = np.random.RandomState(3)
    if args.proto == "flex":
        return flex_main(pkg, torch, C, n, args.iters, rng)
    msgs = [(0x12345, 3, 2, sy.pocsag_alpha_words("THE QUICK BROWN FOX JUMPS OVER THE LAZY DOG 0123456789 " * 3 + "\x04"))] * 12
    bits = sy.pocsag_bits(sy.pocsag_batches(msgs))
    burst = {b: sy.pocsag_pcm(bits, b, noise=900, lead=3000, trail=3000, seed=b) for b in (512, 1200, 2400)}

    def busy(baud, seed):
        x = np.concatenate([burst[baud]] * (n // burst[baud].size + 1))[:n].copy()
        return x

    idle = rng.normal(0, 1500, (C, n)).round().astype(np.int16)
    full = np.stack([busy((512, 1200, 2400)[c % 3], c) for c in range(C)])
    mix = idle.copy()
    mix[::4] = full[::4]
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    for name, host in (("idle", idle), ("busy", full), ("mixed", mix)):
        x = torch.from_numpy(host).to(dev)
        pg = pkg.Pocsag(C, n, device=0)
        for _ in range(3):
            pg.process_device(x.data_ptr(), n, n, stream=stream)
        nev = len(pg.fetch_events())
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(args.iters):
            pg.process_device(x.data_ptr(), n, n, stream=stream)
        t1.record()
        torch.cuda.synchronize()
        ms = t0.elapsed_time(t1) / args.iters
        print(json.dumps({"scenario": name, "channels": C, "pcm_samples_per_channel": n, "ms_per_block": round(ms, 4),
                          "pcm_msamples_per_s": round(C * n / ms / 1e3, 1), "events_last_block": nev,
                          "pcm_read_gbps": round(C * n * 2 / ms / 1e6, 1)}), flush=True)
        pg.close()


def flex_main(pkg, torch, C, n, iters, rng):
    sy = pkg.synth
    recs = [dict(kind="alnum", capcode=1000 + i, text="THE QUICK BROWN FOX JUMPS OVER THE LAZY DOG %d" % i) for i in range(4)]
    frames = []
    for k in range(4):
        ph = {p: sy.flex_phase_words(recs) for p in sy.FLEX_CODINGS[k]["phases"]}
        frames.append(sy.flex_pcm([sy.flex_frame_levels(k, 1, k, ph)], noise=300, seed=k))
    idle = rng.normal(0, 1500, (C, n)).round().astype(np.int16)
    full = np.stack([np.concatenate([frames[c % 4]] * (n // 30000 + 2))[(c * 977) % 30000:][:n] for c in range(C)])
    mix = idle.copy()
    mix[::4] = full[::4]
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    for name, host in (("idle", idle), ("busy", full), ("mixed", mix)):
        x = torch.from_numpy(host).to(dev)
        fx = pkg.Flex(C, n, device=0)
        for _ in range(3):
            fx.process_device(x.data_ptr(), n, n, stream=stream)
        ev, fw = fx.fetch_events()
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(iters):
            fx.process_device(x.data_ptr(), n, n, stream=stream)
        t1.record()
        torch.cuda.synchronize()
        ms = t0.elapsed_time(t1) / iters
        print(json.dumps({"proto": "flex", "scenario": name, "channels": C, "pcm_samples_per_channel": n,
                          "ms_per_block": round(ms, 4), "pcm_msamples_per_s": round(C * n / ms / 1e3, 1),
                          "events_last_block": int(len(ev)), "frames_last_block": int(len(fw)),
                          "pcm_read_gbps": round(C * n * 2 / ms / 1e6, 1)}), flush=True)
        fx.close()


if __name__ == "__main__":
    main()
