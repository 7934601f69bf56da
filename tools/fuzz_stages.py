#!/usr/bin/env python3
"""Randomised GPU-vs-oracle runs of the FLEX stage and of the matrix-core resampler (more shapes and chunkings than the
test-suite has time for).  Exit code 1 on the first difference.

    python tools/fuzz_stages.py [--seconds 120] [--seed 1]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def flex_case(pkg, ora, rng):
    sy = pkg.synth
    C = int(rng.randint(1, 9))
    chans = []
    for c in range(C):
        parts = []
        for _ in range(int(rng.randint(1, 4))):
            k = int(rng.randint(4))
            recs = [dict(kind="alnum", capcode=int(rng.randint(1, 1900000)), text="fuzz %d" % rng.randint(1 << 30)),
                    dict(kind="numeric", capcode=int(rng.randint(1, 1900000)), digits="".join(rng.choice(list("0123456789"), 9)))]
            ph = {p: sy.flex_phase_words(recs[: int(rng.randint(0, 3))]) for p in sy.FLEX_CODINGS[k]["phases"]}
            kw = {}
            r = rng.rand()
            if r < 0.15:
                kw["a_flip"] = int(rng.randint(1, 1 << 16)) << 16
            elif r < 0.3:
                kw["fiw_flip"] = int(rng.randint(1, 1 << 31))
            corrupt = {(p, int(rng.randint(88))): int(rng.randint(1, 1 << 31)) for p in sy.FLEX_CODINGS[k]["phases"]} if rng.rand() < 0.5 else None
            parts.append(sy.flex_frame_levels(k, int(rng.randint(16)), int(rng.randint(128)), ph, corrupt=corrupt, **kw))
        amp = int(rng.choice([300, 3000, 9000, 30000]))
        chans.append(sy.flex_pcm(parts, amplitude=amp, noise=float(rng.choice([0, amp * 0.03, amp * 0.12])), lead=int(rng.randint(0, 4000)),
                                 trail=int(rng.randint(0, 2000)), seed=int(rng.randint(1 << 30)), offset=int(rng.randint(-amp // 4, amp // 4 + 1)),
                                 gap=int(rng.choice([0, 0, 7, 333]))))
    n = max(len(x) for x in chans)
    pcm = np.stack([np.concatenate([x, rng.randint(-200, 201, n - len(x)).astype(np.int16)]) for x in chans])
    max_in = int(rng.choice([n, 70000, 8192, 33333]))
    fx = pkg.binding.Flex(C, max_in)
    got = [[] for _ in range(C)]
    pos = 0
    while pos < n:
        m = min(int(rng.randint(1, max_in + 1)), n - pos)
        ev, fw = fx.process_host(pcm[:, pos:pos + m])
        for e in ev:
            got[int(e["channel"])].append((e, fw[int(e["frame_index"])]["words"].copy() if int(e["type"]) == 1 else None))
        pos += m
    fx.close()
    fields = ("type", "sample", "sync_sample", "coding", "eye", "a", "b", "inv_a", "fiw_raw", "fiw", "fiw_rc", "sample_range",
              "sample_delta", "cycle", "frame")
    nev = 0
    for c in range(C):
        want, _ = ora.Flex().feed(pcm[c])
        g = [tuple(int(e[k]) for k in fields) for e, _ in got[c]]
        w = [tuple(int(e[k]) for k in fields) for e in want]
        if g != w:
            return "flex: events differ on channel %d (%d vs %d)" % (c, len(g), len(w)), 0
        for (e, words), we in zip(got[c], want):
            if words is not None and not np.array_equal(words, we["words"]):
                return "flex: frame words differ on channel %d" % c, 0
        nev += len(w)
    return None, nev


def resampler_case(pkg, ora, rng):
    while True:
        interp, decim = int(rng.randint(1, 17)), int(rng.randint(1, 41))
        if (16 * decim) % interp == 0 and decim * 16 // interp <= 240:
            break
    ntaps = int(rng.randint(max(interp, 4), 900))
    plen = (-(-ntaps // interp) + 3) & ~3
    if -(-decim // interp) > plen:
        return None, 0
    taps = rng.randint(-32639, 32640, ntaps).astype(np.int16)
    if rng.rand() < 0.5:
        taps = (taps // int(rng.choice([2, 64, 700]))).astype(np.int16)
    C = int(rng.randint(1, 6))
    n = int(rng.randint(1000, 120000))
    x = rng.randint(-32768, 32768, (C, n)).astype(np.int16)
    invert = bool(rng.rand() < 0.3)
    max_in = int(rng.choice([n, 65536, 4096]))
    try:
        gpu = pkg.Resampler(C, taps, interp, decim, max_in, device=0, invert=invert)
    except pkg.MfmError:
        return None, 0     # a ratio the stage refuses (LDS), not a parity matter
    refs = [ora.Resampler(taps, interp, decim, invert=invert) for _ in range(C)]
    pos, got, want = 0, [], [[] for _ in range(C)]
    while pos < n:
        m = min(int(rng.randint(1, max_in + 1)), n - pos)
        got.append(gpu.process_host(x[:, pos:pos + m]))
        for c in range(C):
            want[c].append(refs[c].feed(x[c, pos:pos + m]))
        pos += m
    gpu.close()
    got = np.concatenate(got, axis=1)
    want = np.stack([np.concatenate(w) for w in want])
    if got.shape != want.shape or not np.array_equal(got, want):
        return "resampler %d/%d, %d taps, invert %s: outputs differ" % (interp, decim, ntaps, invert), 0
    return None, got.shape[1]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    from __graft_entry__ import load_package
    import oracle_lib as ora
    pkg = load_package()
    rng = np.random.RandomState(args.seed)
    t0 = time.time()
    runs = {"flex": [0, 0], "resampler": [0, 0]}
    while time.time() - t0 < args.seconds:
        for name, fn in (("flex", flex_case), ("resampler", resampler_case)):
            err, units = fn(pkg, ora, rng)
            if err:
                print("FAIL", err, "after", runs)
                return 1
            runs[name][0] += 1
            runs[name][1] += units
    print("ok", {k: {"cases": v[0], "events_or_outputs": v[1]} for k, v in runs.items()})
    return 0


if __name__ == "__main__":
    sys.exit(main())
