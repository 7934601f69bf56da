#!/usr/bin/env python3
"""Round 6: the 128-tap filters on 128-channel slices (mfm_kernel_v3l.hip, two row blocks per wave, whole-tile images) against the
oracle: channel counts around the slice boundaries, ragged blocks, both input formats, forced on and off.  tools/r06/slice128_smoke.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
ora = ge.load_oracle()
b = pkg.binding


def run(tag, plan, nch, n, block, flags=0, u8=False):
    fs, decim, taps, offs, gains = pkg.synth.plan(plan, nr_channels=nch)
    eng = pkg.Engine(fs, decim, block, device=0, flags=flags)
    for o, g in zip(offs, gains):
        eng.add_channel(int(o), taps, float(g))
    eng.commit()
    st = eng.stats()
    cre = np.stack([eng.get_channel(c)[0] for c in range(nch)])
    cim = np.stack([eng.get_channel(c)[1] for c in range(nch)])
    incr = np.stack([eng.get_channel(c)[2] for c in range(nch)])
    iq = pkg.synth.synth_iq(n, fs, list(offs)[:3], seed=nch + block)
    t0 = time.time()
    if u8:
        raw = ((iq.astype(np.int32) >> 8) + 128).clip(0, 255).astype(np.uint8)
        pcm, _ = eng.run_bytes(raw, block, b.MFM_IN_RTLSDR_U8)
        iq = ((raw.astype(np.int16) - 127) << 7).astype(np.int16)   # multifm/rtl_sdr_if.c:146-148
    else:
        pcm, _ = eng.run(iq, block)
    ref, _ = ora.run_channels(iq, cre, cim, incr, decim, threads=8)
    eng.close()
    ok = pcm.shape == ref.shape and np.array_equal(pcm, ref)
    bad = int((pcm != ref).sum()) if pcm.shape == ref.shape else -1
    first = tuple(np.argwhere(pcm != ref)[0]) if bad > 0 else None
    print(f"{tag:28s} C={nch:5d} block={block:7d} variant={st['kernel_variant']} ksteps={st['k_steps']} lds={st['lds_bytes']:6d} "
          f"{'OK ' if ok else 'FAIL'} bad={bad} first={first} shape={pcm.shape} {time.time() - t0:.1f}s", flush=True)
    return ok


def main():
    ok = True
    for nch in (128, 129, 130, 200, 256, 257, 1024, 65, 16):
        n = 96 * (900 if nch < 512 else 300) + 128 + 5
        for block in (1 << 17, 30001):
            ok &= run("slice128", "cfg3_1024ch", nch, n, block, flags=b.MFM_F_SLICE_128)
        ok &= run("default", "cfg3_1024ch", nch, n, 1 << 17)
        ok &= run("slice64", "cfg3_1024ch", nch, n, 1 << 17, flags=b.MFM_F_SLICE_64)
        if hasattr(pkg.Engine, "run_bytes"):
            ok &= run("slice128/u8", "cfg3_1024ch", nch, n, 1 << 17, flags=b.MFM_F_SLICE_128, u8=True)
    ok &= run("slice128 grid", "cfg2_64ch", 192, 96 * 900 + 133, 1 << 16, flags=b.MFM_F_SLICE_128)
    print("ALL OK" if ok else "FAILURES")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
