#!/usr/bin/env python3
"""Round 5: time of the DC blocker pass (filter/dc_blocker.h:71-93 on the GPU, csrc/mfm_resampler.hip) on the PCM of one
2^26-sample block of 64 channels (699 050 samples per channel in, 16/25 resampled: 447 392 out), against the oracle on a
short run.  tools/r05/dc_time.py"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
ora = ge.load_oracle()
nch, n_in = 64, (1 << 26) // 96
taps = (np.hanning(821) / 821 * 16384 * 16).astype(np.int16)
rng = np.random.RandomState(3)
small = rng.randint(-20000, 20000, size=(nch, 40000)).astype(np.int16)
for dc in (None, 0.9999):
    gpu = pkg.Resampler(nch, taps, 16, 25, n_in, device=0, dc_pole=dc)
    refs = [ora.Resampler(taps, 16, 25, dc_pole=dc) for _ in range(nch)]
    got = gpu.process_host(small)
    want = np.stack([r.feed(small[c]) for c, r in enumerate(refs)])
    ok = got.shape == want.shape and np.array_equal(got, want)
    x = torch.from_numpy(rng.randint(-20000, 20000, size=(nch, n_in)).astype(np.int16)).cuda()
    torch.cuda.synchronize()
    for _ in range(3):
        gpu.process_device(x.data_ptr(), n_in, n_in)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        gpu.process_device(x.data_ptr(), n_in, n_in)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    print(f"dc_pole={dc}: parity {'OK' if ok else 'FAIL'}; resampler{' + DC blocker' if dc else ''} {ms:.3f} ms per block of {nch} x {n_in} samples", flush=True)
    gpu.close()
