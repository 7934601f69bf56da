"""Kernel time of the channel engine on 8-bit input (mfm_engine_push_bytes): the block read as bytes by the matrix
kernel (default) against widened to int16 in HBM first (--widen: MFM_F_WIDEN_8BIT, the round-1 path), and against an
int16 block of the same length (--fmt 0).  One JSON line; the kernel's duration is MFM_F_TIMING's (HIP events on the
compute stream), the widening pass shows in a rocprofv3 kernel trace of the same command (tools/prof_ingest8.sh).

    python tools/bench_ingest8.py [--fmt 3] [--widen] [--block-log2 26] [--steps 12] [--config cfg2_64ch]

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
    ap.add_argument("--fmt", type=int, default=3, help="MFM_IN_*: 0 cs16, 1 cs8, 2 cu8, 3 rtlsdr u8")
    ap.add_argument("--widen", action="store_true")
    ap.add_argument("--block-log2", type=int, default=26)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="cfg2_64ch")
    ap.add_argument("--channels", type=int, default=0)
    args = ap.parse_args()
    from __graft_entry__ import load_package
    pkg = load_package()
    b = pkg.binding
    kw = {"nr_channels": args.channels} if args.channels else {}
    fs, decim, taps, offs, gains = pkg.synth.plan(args.config, **kw)
    block = 1 << args.block_log2
    flags = b.MFM_F_DEVICE_ONLY | b.MFM_F_TIMING | (b.MFM_F_WIDEN_8BIT if args.widen else 0)
    eng = pkg.Engine(fs, decim, block, device=0, flags=flags)
    for o, g in zip(offs, gains):
        eng.add_channel(int(o), taps, float(g))
    eng.commit()
    rng = np.random.RandomState(5)
    if args.fmt == 0:
        base = pkg.synth.synth_iq(1 << 22, fs, offs[:: max(1, len(offs) // 8)][:8], seed=7).reshape(-1, 2)
        data = np.tile(base, (block // base.shape[0] + 1, 1))[:block]
    else:
        # an RTL-SDR-like capture: the synthetic wideband signal scaled to 8 bits around mid-scale
        base = pkg.synth.synth_iq(1 << 22, fs, offs[:: max(1, len(offs) // 8)][:8], seed=7).reshape(-1, 2)
        v = np.clip((base.astype(np.int32) >> 7) + (127 if args.fmt == 3 else 0), -128 if args.fmt != 3 else 0,
                    127 if args.fmt != 3 else 255)
        data = np.tile(v.astype(np.uint8 if args.fmt == 3 else np.int8).view(np.uint8), (block // base.shape[0] + 1, 1))[:block]
    for k in range(args.warmup + args.steps):
        rc = eng.push(data.reshape(-1)) if args.fmt == 0 else eng.push_bytes(data, args.fmt)
        assert rc == 0, rc
    eng.sync()
    ms = eng.launch_ms()[-args.steps:]
    st = eng.stats()
    outs = block // decim
    alg = block * (4 if args.fmt == 0 else 2) + len(offs) * outs * 2
    avg = float(np.mean(ms))
    print(json.dumps({"input": ["cs16", "cs8", "cu8", "rtlsdr_u8"][args.fmt], "read_as": "int16 (widened in HBM first)" if
                      (args.widen and args.fmt) else ("int16" if args.fmt == 0 else "bytes"), "channels": len(offs),
                      "block_samples": block, "kernel_ms_per_block": round(avg, 4), "kernel_ms_min": round(float(np.min(ms)), 4),
                      "msamples_per_s": round(block / avg / 1e3, 1), "algorithmic_bytes": alg,
                      "hbm_gbps": round(alg / avg / 1e6, 1), "launches": st["launches"], "launches_8bit": st["launches_8bit"],
                      "kernel_variant": st["kernel_variant"]}), flush=True)
    eng.close()


if __name__ == "__main__":
    main()
