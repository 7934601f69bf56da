#!/usr/bin/env python3
"""Floating-point IQ path against the integer path on the same shape (BASELINE configs[4] "fp32 vs int16 IQ path"):
kernel time per block from HIP events on the launch stream, input resident in HBM.

    python tools/bench_f32.py [--config cfg2_64ch|cfg5_airspy] [--channels N] [--block-log2 24] [--iters 20]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cfg2_64ch")
    ap.add_argument("--channels", type=int, default=64)
    ap.add_argument("--block-log2", type=int, default=24)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--tile-kernel", action="store_true", help="the round-1 kernel (MFM_F32_TILE_KERNEL)")
    args = ap.parse_args()
    import torch
    from __graft_entry__ import load_package
    pkg = load_package()
    fs, decim, taps, offs, gains = pkg.synth.plan(args.config, nr_channels=args.channels)
    blk = 1 << args.block_log2
    T = len(taps)
    base = pkg.synth.synth_iq(1 << 20, fs, offs[:: max(1, len(offs) // 8)][:8], seed=7)
    iq16 = np.tile(base, (blk // base.shape[0] + 1, 1))[:blk]
    d_f = torch.from_numpy(iq16.astype(np.float32).reshape(-1)).cuda()
    eng = pkg.F32Engine(fs, decim, blk, device=0, tile_kernel=args.tile_kernel)
    for o, g in zip(offs, gains):
        eng.add_channel(int(o), taps, float(g))
    eng.commit()
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(5):
        b = eng.process_device(d_f.data_ptr(), blk, stream=st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters):
        b = eng.process_device(d_f.data_ptr(), blk, stream=st)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.iters
    C = len(offs)
    nout = b.nr_out
    flops = 8.0 * C * T * nout                      # 4 FMAs per complex tap
    bytes_alg = blk * 8 + C * nout * 6              # float IQ in, float + int16 PCM out
    line = {"path": "fp32 IQ", "config": args.config, "channels": C, "taps": T, "decimation": decim,
            "block_samples": blk, "ms_per_block": round(ms, 4),
            "msamp_per_s_x_channels": round(blk * C / ms / 1e3, 1),
            "fp32_tflops": round(flops / ms / 1e9, 2), "fp32_vector_peak_tflops": 157.3,
            "hbm_algorithmic_gbps": round(bytes_alg / ms / 1e6, 1)}
    # the integer engine on the same shape
    d_i = torch.from_numpy(iq16.reshape(-1)).cuda()
    lib = pkg.load_library()
    in_bytes = lib.mfm_engine_input_bytes(blk, T)
    bufs = [torch.zeros(in_bytes // 2, dtype=torch.int16, device="cuda") for _ in range(2)]
    ie = pkg.Engine(fs, decim, blk, device=0, flags=pkg.binding.MFM_F_DEVICE_ONLY | pkg.binding.MFM_F_TIMING,
                    ext_input=(bufs[0].data_ptr(), bufs[1].data_ptr()))
    for o, g in zip(offs, gains):
        ie.add_channel(int(o), taps, float(g))
    ie.commit()
    for b_ in bufs:
        b_[: 2 * blk].copy_(d_i) if in_bytes // 2 >= 2 * blk else None
    for _ in range(5):
        ie.acquire_input()
        ie.submit(blk)
    ie.sync()
    s0 = ie.stats()
    for _ in range(args.iters):
        ie.acquire_input()
        ie.submit(blk)
    ie.sync()
    s1 = ie.stats()
    ims = (s1["kernel_ms"] - s0["kernel_ms"]) / max(1, s1["launches"] - s0["launches"])
    line["int16_path_ms_per_block"] = round(ims, 4)
    line["int16_kernel"] = "mfma" if s1["kernel_variant"] == 1 else "dot2"
    line["fp32_over_int16_time"] = round(ms / ims, 2)
    print(json.dumps(line))
    eng.close()
    ie.close()


if __name__ == "__main__":
    main()
