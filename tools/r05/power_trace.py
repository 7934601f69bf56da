#!/usr/bin/env python3
"""Round 5: board power and clock of the card the run uses (matched by PCI address, bench.gpu_sysfs_sample) while ONE shape runs
back to back for several seconds - is the sustained clock the power cap's?  tools/r05/power_trace.py [seconds] [plan:channels ...]
Prints one line per sample: t, sclk MHz (sysfs, instantaneous), PPT watts (sysfs), and per shape the clock inside the launches by
the kernel's own stamps over the last launches."""
import json
import os
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402
import bench  # noqa: E402

pkg = ge.load_package()
b = pkg.binding
lib = pkg.load_library()


def run(plan, nch, seconds, block=1 << 26):
    fs, decim, taps, offs, gains = pkg.synth.plan(plan, nr_channels=nch)
    in_bytes = lib.mfm_engine_input_bytes(block, len(taps))
    bufs = [torch.empty(in_bytes // 2, dtype=torch.int16, device="cuda") for _ in range(2)]
    eng = pkg.Engine(fs, decim, block, device=0, flags=b.MFM_F_DEVICE_ONLY | b.MFM_F_TIMING,
                     ext_input=(bufs[0].data_ptr(), bufs[1].data_ptr()))
    for o, g in zip(offs, gains):
        eng.add_channel(int(o), taps, float(g))
    eng.commit()
    base = pkg.synth.synth_iq(1 << 22, fs, offs[:: max(1, len(offs) // 8)][:8], seed=11)
    host = np.tile(base, (-(-(in_bytes // 4) // base.shape[0]), 1))[: in_bytes // 4].reshape(-1)
    for t in bufs:
        t.copy_(torch.from_numpy(host))
    torch.cuda.synchronize()
    addr = bench.device_pci_address(0)
    samples, stop = [], threading.Event()

    def sampler():
        t0 = time.perf_counter()
        while not stop.is_set():
            s = bench.gpu_sysfs_sample(0, addr)
            if s:
                samples.append((time.perf_counter() - t0, s["sclk_mhz"], s["power_w"], s["power_cap_w"], s["matched"]))
            time.sleep(0.25)
    th = threading.Thread(target=sampler, daemon=True)
    th.start()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(16):
            eng.acquire_input()
            eng.submit(block, producer_stream=0, wait_producer=False)
            n += 1
        eng.sync()
    stop.set()
    th.join()
    ms = float(np.mean(eng.launch_ms(16)))
    cyc = bench.launch_clocks(eng, 16)
    eng.close()
    print(f"## {plan} x {nch} channels: {n} launches in {seconds:.0f} s, kernel {ms:.4f} ms (last 16), clocks {json.dumps(cyc)[:260]}")
    for s in samples:
        print("   t %5.2f s  sclk %6.0f MHz  PPT %6.0f W  cap %5.0f W  matched %s" % s)


if __name__ == "__main__":
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
    shapes = sys.argv[2:] or ["cfg2_64ch:64", "cfg5_airspy:256", "cfg3_1024ch:1024"]
    for sh in shapes:
        p, c = sh.split(":")
        run(p, int(c), secs)
        time.sleep(2.0)
