#!/usr/bin/env python3
"""Device-resident FLEX chain at the headline geometry: wideband int16 IQ (resident in HBM, 2.4 MS/s) -> channel engine
(D = 96, 128 taps, 25 kS/s per channel) -> 16/25 resampler (the reference's decoder runs FLEX behind -I 16 -D 25 with an
821-tap filter, etc/resampler_filter.json) -> FLEX stage.  Everything is queued on the engine's stream; per-stage times from
events on that stream; the only host traffic is the event list of the last block.

    python tools/bench_chain_dev.py [--channels 64] [--block-log2 26] [--iters 20] [--busy]

--busy: every channel carries back-to-back FLEX frames (FM carriers synthesised once for 2^21 samples and tiled - frames
repeat, so sync and the frame gather run on every channel); default: eight FM carriers with a tone, i.e. the FLEX stage
mostly searches.  Not part of bench.py's contract line."""
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
    ap.add_argument("--block-log2", type=int, default=26)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--resampler-taps", type=int, default=821)
    ap.add_argument("--resampler-dot2", action="store_true", help="the v_dot2 resampler kernel (MFM_RS_FORCE_DOT2)")
    ap.add_argument("--proto", default="flex", choices=["flex", "pocsag"],
                    help="pocsag: etc/pocsag_rtlsdr.json geometry (1.2 MS/s, D = 25 -> 48 kS/s) -> 4/5 (81 taps) -> POCSAG stage")
    ap.add_argument("--dc-block", action="store_true", help="decoder -b: the DC blocker behind the resampler (pole 0.9999)")
    ap.add_argument("--in8", action="store_true", help="the wideband block is RTL-SDR bytes (mfm_engine_acquire_input_bytes), "
                    "read as they are by the matrix kernel")
    args = ap.parse_args()
    import torch
    from __graft_entry__ import load_package
    pkg = load_package()
    sy = pkg.synth
    C = args.channels
    if args.proto == "pocsag":
        fs, decim, taps, offs, gains = sy.plan("pocsag_rtlsdr", nr_channels=C)
    else:
        fs, decim, taps, offs, gains = sy.plan("cfg2_64ch" if C <= 64 else "cfg3_1024ch", nr_channels=C)
    block = 1 << args.block_log2
    lib = pkg.load_library()
    in_bytes = lib.mfm_engine_input_bytes(block, len(taps))
    bufs = [torch.empty(in_bytes // 2, dtype=torch.int16, device="cuda") for _ in range(2)]
    eng = pkg.Engine(fs, decim, block, device=0, flags=pkg.binding.MFM_F_DEVICE_ONLY,
                     ext_input=(bufs[0].data_ptr(), bufs[1].data_ptr()))
    for o, g in zip(offs, gains):
        eng.add_channel(int(o), taps, float(g))
    eng.commit()
    base = sy.synth_iq(1 << 22, fs, offs[:: max(1, C // 8)][:8], seed=7)
    host = np.tile(base, (-(-(in_bytes // 4) // base.shape[0]), 1))[: in_bytes // 4].reshape(-1)
    if args.in8:
        # the same signal as an RTL-SDR would deliver it: 8 bits around mid-scale, two bytes per sample, all over the buffers
        u8 = np.clip((host.astype(np.int32) >> 7) + 127, 0, 255).astype(np.uint8)
        host = np.concatenate([u8, u8])[: in_bytes].view(np.int16)
    for b in bufs:
        b.copy_(torch.from_numpy(host))
    ri, rd, rn = (4, 5, 81) if args.proto == "pocsag" else (16, 25, args.resampler_taps)
    rt = sy.design_lpf(rn, 0.45 / max(ri, rd), 1.0) * ri
    rtaps = np.array([int(t * 16384.0) for t in rt], dtype=np.int16)
    rs = pkg.Resampler(C, rtaps, ri, rd, block // decim + 8, device=0, force_dot2=args.resampler_dot2,
                       dc_pole=0.9999 if args.dc_block else None)
    fx = pkg.Pocsag(C, rs.max_out(), device=0) if args.proto == "pocsag" else pkg.Flex(C, rs.max_out(), device=0)
    st = torch.cuda.ExternalStream(eng.stream)
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(args.iters)]

    def step(marks):
        if args.in8:
            eng.acquire_input_bytes(pkg.binding.MFM_IN_RTLSDR_U8)
        else:
            eng.acquire_input()
        if marks:
            marks[0].record(st)
        eng.submit(block, producer_stream=0, wait_producer=False)
        dptr, stride, nout, _ = eng.last_output_device()
        if marks:
            marks[1].record(st)
        yptr, ystride, ny = rs.process_device(dptr, stride, nout, stream=eng.stream)
        if marks:
            marks[2].record(st)
        fx.process_device(yptr, ystride, ny, stream=eng.stream)
        if marks:
            marks[3].record(st)
        return nout, ny

    for _ in range(60):
        step(None)
    eng.sync()
    nout = ny = 0
    for i in range(args.iters):
        nout, ny = step(ev[i])
    eng.sync()
    torch.cuda.synchronize()
    events = fx.fetch_events()
    events = events[0] if isinstance(events, tuple) else events
    t = np.array([[e[0].elapsed_time(e[k]) for k in (1, 2, 3)] for e in ev])
    stage = np.diff(np.concatenate([np.zeros((args.iters, 1)), t], 1), axis=1)
    med = np.median(stage, 0)
    total = float(np.median(t[:, 2]))
    print(json.dumps({"input": "rtl-sdr u8 bytes" if args.in8 else "int16", "launches_8bit": eng.stats()["launches_8bit"],
                      "chain": "IQ (HBM) -> engine (D %d, %d taps) -> resampler %d/%d (%d taps) -> %s stage" % (decim, len(taps), ri, rd, rn, args.proto.upper()),
                      "channels": C, "block_samples": block, "pcm_in_per_channel": int(nout), "pcm_out_per_channel": int(ny),
                      "ms_engine": round(float(med[0]), 4), "ms_resampler": round(float(med[1]), 4), "ms_pager": round(float(med[2]), 4),
                      "ms_per_block": round(total, 4), "msamp_per_s_x_channels": round(block * C / total / 1e3, 1),
                      "events_last_block": int(len(events)), "kernel": eng.stats()["kernel_variant"]}))
    eng.close()
    rs.close()
    fx.close()


if __name__ == "__main__":
    main()
