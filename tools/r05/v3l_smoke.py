#!/usr/bin/env python3
"""Round 5: first contact of the long-filter kernel (mfm_kernel_v3l.hip) with the GPU - a handful of geometries against
the oracle, with what the engine selected.  tools/r05/v3l_smoke.py [quick]"""
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


def run(tag, fs, decim, taps, offs, n, block, gains=None, flags=0, fmt=None):
    gains = gains if gains is not None else [1.0] * len(offs)
    eng = pkg.Engine(fs, decim, block, device=0, flags=flags)
    for o, g in zip(offs, gains):
        eng.add_channel(int(o), taps, float(g))
    eng.commit()
    st = eng.stats()
    nch = len(offs)
    cre = np.stack([eng.get_channel(c)[0] for c in range(nch)])
    cim = np.stack([eng.get_channel(c)[1] for c in range(nch)])
    incr = np.stack([eng.get_channel(c)[2] for c in range(nch)])
    iq = pkg.synth.synth_iq(n, fs, list(offs)[:3], seed=decim + len(taps))
    t0 = time.time()
    pcm, _ = eng.run(iq, block)
    ref, _ = ora.run_channels(iq, cre, cim, incr, decim, threads=8)
    eng.close()
    ok = pcm.shape == ref.shape and np.array_equal(pcm, ref)
    bad = int((pcm != ref).sum()) if pcm.shape == ref.shape else -1
    first = tuple(np.argwhere(pcm != ref)[0]) if bad > 0 else None
    print(f"{tag:34s} D={decim:4d} T={len(taps):4d} C={nch:4d} variant={st['kernel_variant']} ksteps={st['k_steps']} "
          f"mask={st['tap_hi_mask']:#06x} lds={st['lds_bytes']:6d} {'OK ' if ok else 'FAIL'} bad={bad} first={first} "
          f"shape={pcm.shape} {time.time() - t0:.1f}s", flush=True)
    return ok


def main():
    ok = True
    fs = 4000000
    offs = [25000 * k + (137 if k % 3 == 0 else 0) for k in range(-10, 11)]
    for decim, ntaps in [(96, 512), (96, 256), (120, 512), (100, 256), (25, 256), (400, 512), (200, 256), (48, 129), (100, 400),
                         (320, 512), (448, 512)]:
        for shape, scale in (("lpf", 1.0), ("lpf_div8", 0.125), ("lpf_x8", 8.0)):
            taps = pkg.synth.design_lpf(ntaps, 12500.0, fs) * scale
            for block in (1 << 16, 30001):
                ok &= run(f"{shape}/{block}", fs, decim, taps, offs, decim * 400 + ntaps + 7, block)
        if len(sys.argv) > 1 and sys.argv[1] == "quick":
            break
    # the bench's shapes at a small size
    for plan, nch in (("cfg5_airspy", 256), ("cfg5_airspy", 130), ("cfg5_airspy", 77), ("pocsag_rtlsdr_256taps", 64), ("pocsag_rtlsdr_256taps", 200),
                      ("pocsag_airspy", 64), ("pocsag_airspy", 129), ("multifm_airspy", 64), ("multifm_airspy", 192)):
        fs2, decim, taps, offs2, gains = pkg.synth.plan(plan, nr_channels=nch)
        ok &= run(plan, fs2, decim, taps, offs2, decim * 700 + len(taps) + 3, 1 << 17, gains=gains)
        ok &= run(plan + "/30001", fs2, decim, taps, offs2, decim * 700 + len(taps) + 3, 30001, gains=gains)
        ok &= run(plan + "/one-rb", fs2, decim, taps, offs2, decim * 700 + len(taps) + 3, 1 << 17, gains=gains, flags=b.MFM_F_V3L_ONE_ROW_BLOCK)
    print("ALL OK" if ok else "FAILURES")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
