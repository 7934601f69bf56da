#!/usr/bin/env python3
"""round 6: shapes the fuzzers found (small decimations on the scheduled long-filter kernel): where do they differ from the oracle?"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
ora = ge.load_oracle()
b = pkg.binding


def run(tag, fs, decim, ntaps, nch, n, block, flags=0, want_iq=False):
    taps = pkg.synth.design_lpf(ntaps, 12500.0, fs)
    rng = np.random.RandomState(nch + decim)
    offs = rng.randint(-fs // 2, fs // 2, size=nch)
    eng = pkg.Engine(fs, decim, block, device=0, flags=flags)
    for o in offs:
        eng.add_channel(int(o), taps, 1.0, want_iq=want_iq)
    eng.commit()
    st = eng.stats()
    cre = np.stack([eng.get_channel(c)[0] for c in range(nch)])
    cim = np.stack([eng.get_channel(c)[1] for c in range(nch)])
    incr = np.stack([eng.get_channel(c)[2] for c in range(nch)])
    iq = pkg.synth.random_iq(n, seed=nch)
    pcm, q = eng.run(iq, block)
    ref, refq = ora.run_channels(iq, cre, cim, incr, decim, threads=8, want_iq=want_iq)
    eng.close()
    ok = pcm.shape == ref.shape and np.array_equal(pcm, ref)
    msg = ""
    if not ok and pcm.shape == ref.shape:
        bad = np.argwhere(pcm != ref)
        chans = np.unique(bad[:, 0])
        outs = np.unique(bad[:, 1])
        msg = f" bad={len(bad)} channels {chans[:8]}..{chans[-1]} ({len(chans)}) outputs {outs[:12]}..{outs[-1]} ({len(outs)}) mod64 {np.unique(outs % 64)[:20]}"
    okq = True
    if want_iq and q is not None:
        okq = np.array_equal(q, refq)
    print(f"{tag:20s} D={decim:3d} T={ntaps:3d} C={nch:3d} n={n} block={block} variant={st['kernel_variant']} ksteps={st['k_steps']} lds={st['lds_bytes']} "
          f"{'OK' if ok else 'FAIL'} iq {'OK' if okq else 'FAIL'}{msg}", flush=True)
    return ok


# split rows (decimations that are not multiples of 4): one and four staging chunks per thread, both formats via want_iq on / off
for (fs, d, t, c, n, blk, iq) in ((1200000, 25, 256, 64, 200000, 65536, False), (1200000, 25, 256, 200, 200000, 30001, True), (1200000, 25, 400, 9, 120000, 4096, False),
                                  (2400000, 50, 200, 70, 300000, 100000, False), (2400000, 75, 300, 130, 400000, 65536, True), (2400000, 33, 200, 17, 200000, 7001, False),
                                  (2400000, 110, 512, 64, 500000, 131072, False), (2400000, 125, 256, 64, 500000, 131072, False), (1000000, 9, 300, 5, 90000, 30000, True)):
    run("split", fs, d, t, c, n, blk, 0, want_iq=iq)
for flags, tag in ((0, "default"), (b.MFM_F_SLICE_128, "slice128"), (b.MFM_F_V3L_ONE_ROW_BLOCK, "one-rb")):
    run(tag, 2400000, 16, 300, 130, 90000, 100000, flags)
    run(tag, 2400000, 8, 512, 130, 320000, 90001, flags)
    run(tag, 2400000, 8, 512, 130, 320000, 30001, flags)
    run(tag, 2400000, 8, 512, 200, 320000, 90001, flags)
    run(tag, 10000000, 32, 256, 130, 741019, 100000, flags, want_iq=True)
    run(tag, 10000000, 32, 256, 130, 741019, 33333, flags, want_iq=True)
    run(tag, 1200000, 32, 100, 257, 367494, 367494, flags)
    run(tag, 2400000, 96, 512, 130, 700000, 100000, flags)
    run(tag, 10000000, 400, 512, 256, 2000000, 262144, flags)
