#!/usr/bin/env python3
"""Round 5: the long-filter shapes of bench.py's other_geometries, second-generation long-filter kernel (auto) against the
first generation (MFM_F_FORCE_MFMA_V1) on one box, same protocol as bench.py (blocks resident in HBM, settle phase, kernel
duration from the engine's HIP events).  tools/r05/long_ab.py [log2 block] [8bit]"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
b = pkg.binding
lib = pkg.load_library()

SHAPES = (("configs4_int16_share", "cfg5_airspy", 256), ("pocsag_rtlsdr_d25_256taps", "pocsag_rtlsdr_256taps", 64),
          ("pocsag_airspy_d100_256taps", "pocsag_airspy", 64), ("multifm_airspy_d120_512taps", "multifm_airspy", 64),
          ("cfg2_512taps", "cfg2_64ch:512", 64), ("cfg2_256taps", "cfg2_64ch:256", 64))


def one(plan, nch, block, flags, steps=24, settle_s=0.25):
    ntaps = None
    if ":" in plan:
        plan, ntaps = plan.split(":")
    fs, decim, taps, offs, gains = pkg.synth.plan(plan, nr_channels=nch)
    if ntaps:
        taps = pkg.synth.design_lpf(int(ntaps), 12500.0, fs)
    in_bytes = lib.mfm_engine_input_bytes(block, len(taps))
    bufs = [torch.empty(in_bytes // 2, dtype=torch.int16, device="cuda") for _ in range(2)]
    eng = pkg.Engine(fs, decim, block, device=0, flags=b.MFM_F_DEVICE_ONLY | b.MFM_F_TIMING | flags,
                     ext_input=(bufs[0].data_ptr(), bufs[1].data_ptr()))
    for o, g in zip(offs, gains):
        eng.add_channel(int(o), taps, float(g))
    eng.commit()
    base = pkg.synth.synth_iq(1 << 22, fs, offs[:: max(1, len(offs) // 8)][:8], seed=11)
    host = np.tile(base, (-(-(in_bytes // 4) // base.shape[0]), 1))[: in_bytes // 4].reshape(-1)
    for t in bufs:
        t.copy_(torch.from_numpy(host))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < settle_s:
        for _ in range(8):
            eng.acquire_input()
            eng.submit(block, producer_stream=0, wait_producer=False)
        eng.sync()
    for _ in range(steps):
        eng.acquire_input()
        eng.submit(block, producer_stream=0, wait_producer=False)
    eng.sync()
    ms = float(np.mean(eng.launch_ms(steps)))
    st = eng.stats()
    eng.close()
    return {"decimation": decim, "taps": len(taps), "channels": nch, "kernel_ms": ms, "variant": st["kernel_variant"],
            "k_steps": st["k_steps"], "mask": st["tap_hi_mask"], "lds": st["lds_bytes"],
            "ps_per_chan_out": ms * 1e9 / (nch * (block // decim))}


def main():
    block = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 26)
    out = {}
    for key, plan, nch in SHAPES:
        row = {}
        for name, flags in (("v3l", 0), ("v1", b.MFM_F_FORCE_MFMA_V1), ("v3l_again", b.MFM_F_V3L_ONE_ROW_BLOCK)):
            try:
                row[name] = one(plan, nch, block, flags)
            except Exception as e:
                row[name] = {"error": repr(e)}
        out[key] = row
        a, c = row.get("v3l", {}), row.get("v1", {})
        if "kernel_ms" in a and "kernel_ms" in c:
            print(f"{key:32s} D={a['decimation']:4d} T={a['taps']:4d} C={nch:4d} mask={a['mask']:#06x} v3l {a['kernel_ms']:.4f} ms "
                  f"({a['ps_per_chan_out']:.2f} ps, variant {a['variant']}, lds {a['lds']})  v1 {c['kernel_ms']:.4f} ms "
                  f"({c['ps_per_chan_out']:.2f} ps)  ratio {a['kernel_ms'] / c['kernel_ms']:.3f}  one-row-block {row['v3l_again'].get('kernel_ms', 0):.4f}",
                  flush=True)
        else:
            print(key, row, flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
