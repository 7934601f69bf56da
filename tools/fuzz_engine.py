#!/usr/bin/env python3
"""Randomised GPU-vs-oracle runs of the channel engine: random sample rate / decimation / filter length / channel set /
gains / block sizes / kernel selection, random full-scale or FM input; PCM (and, half the time, the filtered IQ) must be
bit-exact.  Exit code 1 on the first difference.

    python tools/fuzz_engine.py [--seconds 300] [--seed 1] [--ingest8 | --stream] [--long]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def case(pkg, ora, rng, long_only=False, slice128=False):
    b = pkg.binding
    decim = int(rng.choice([25, 32, 40, 40, 64, 96, 96, 96, 100, 128, 400, 8 * int(rng.randint(1, 16)), int(rng.randint(6, 200))]))
    fs = int(rng.choice([1000000, 1200000, 2400000, 10000000]))
    ntaps = int(rng.choice([decim, 128, 128, 129, 200, 256, 512, decim + int(rng.randint(0, 100))]))
    ntaps = max(ntaps, decim)
    if ntaps > 600:
        ntaps = 512 if decim <= 512 else decim
    if long_only:
        # filters of 129..512 taps at any decimation up to 460: the resident and the streamed instances of the first-generation
        # matrix kernel over their staging-chunk counts, tile forms and tap-plane masks
        decim = int(rng.choice([int(rng.randint(8, 461)), 4 * int(rng.randint(2, 116)), 8 * int(rng.randint(1, 58)), 96, 400]))
        ntaps = max(decim, int(rng.choice([int(rng.randint(129, 513)), 129, 160, 256, 257, 400, 512])))
    small = (not long_only) and rng.rand() < 0.12
    if small:
        # decimations 1 .. 7: 1, 2, 4 run the shifted-copies form of the long-filter kernel (etc/multifm_file.json channelises
        # without decimating), the others the first generation or v_dot2
        decim = int(rng.choice([1, 1, 2, 4, 4, 3, 5, 6, 7]))
        ntaps = int(rng.choice([128, 128, 64, 65, 200, 256, 512, 16, decim + int(rng.randint(0, 300))]))
    nch = int(rng.choice([1, 2, 5, 8, 9, 16, 33, 64, 65, 130, int(rng.randint(1, 300))]))
    if slice128:
        # round 6: filters of up to 128 taps at decimations that are multiples of 32 on SLICES OF 128 CHANNELS (two row blocks per
        # wave, whole-tile images of mfm_kernel_v3l.hip): channel counts around the slice and row-block boundaries
        decim = int(rng.choice([96, 96, 96, 64, 128, 32]))
        ntaps = int(rng.choice([128, 128, 128, 127, 100, decim, decim + int(rng.randint(0, 129 - decim))]))
        ntaps = max(decim, min(ntaps, 128))
        nch = int(rng.choice([128, 129, 130, 136, 200, 255, 256, 257, 300, 65, 17, int(rng.randint(1, 400))]))
        small = False
    taps = pkg.synth.design_lpf(ntaps, float(rng.choice([5000.0, 12500.0, 40000.0])), fs) * float(rng.choice([1.0, 1.0, 3.0, 0.2]))
    if long_only and rng.rand() < 0.25:
        taps = rng.uniform(-0.3, 0.3, ntaps)  # high bytes in every k-step: the all-planes instances
    offs = rng.randint(-fs // 2, fs // 2, size=nch)
    offs[: min(nch, 4)] = [0, 25000, -37500, 3125][: min(nch, 4)]
    if fs % (4 * decim) == 0 and rng.rand() < 0.6:
        # channels on the raster of a quarter of the output rate: exact rotators (identity, sign flip, quarter turns), all of
        # them or a random share, so that launches, slices and waves of every class mix occur
        q = fs // (4 * decim)
        snap = rng.rand(nch) < float(rng.choice([1.0, 0.9, 0.5]))
        offs = np.where(snap, q * rng.randint(-2 * decim + 1, 2 * decim, size=nch), offs)
    gains = rng.choice([1.0, 2.5118864315095806, 0.3], size=nch)
    kernel = str(rng.choice(["auto", "auto", "mfma1", "dot2"]))
    want_iq = bool(rng.rand() < 0.5)
    stream = bool(rng.rand() < 0.2)
    if long_only:
        kernel, want_iq = "auto", bool(rng.rand() < 0.3)
    if slice128:
        kernel, want_iq = "auto", bool(rng.rand() < 0.25)
    n = int(rng.randint(ntaps, 400000))
    if slice128:
        n = int(rng.randint(ntaps, max(ntaps + 1, min(400000, int(4.0e9 * decim / (ntaps * nch))))))
    if small:
        # bounded oracle work: n / D outputs x taps x channels
        n = int(rng.randint(ntaps, max(ntaps + 1, min(400000, int(2.0e9 * decim / (ntaps * nch))))))
    if rng.rand() < 0.5:
        iq = pkg.synth.random_iq(n, seed=int(rng.randint(1 << 30)))
    else:
        iq = pkg.synth.synth_iq(n, fs, offs[: min(nch, 6)], seed=int(rng.randint(1 << 30)))
    max_block = int(rng.choice([n, 65536, 8192, 100000]))
    flags = (b.MFM_F_FORCE_DOT2 if kernel == "dot2" else 0) | (b.MFM_F_FORCE_MFMA_V1 if kernel == "mfma1" else 0) | \
            (b.MFM_F_STREAM_TAPS if stream else 0) | (b.MFM_F_V3L_ONE_ROW_BLOCK if rng.rand() < 0.3 else 0) | \
            (b.MFM_F_SLICE_128 if (slice128 and rng.rand() < 0.9) or rng.rand() < 0.3 else 0)
    try:
        eng = pkg.Engine(fs, decim, max_block, device=0, flags=flags)
        for o, g in zip(offs, gains):
            eng.add_channel(int(o), taps, float(g), want_iq=want_iq)
        eng.commit()
    except pkg.MfmError as e:
        return None, "refused: %s" % e
    variant = eng.stats()["kernel_variant"]
    resident = eng.stats()["taps_resident"]
    cre = np.stack([eng.get_channel(c)[0] for c in range(nch)])
    cim = np.stack([eng.get_channel(c)[1] for c in range(nch)])
    incr = np.stack([eng.get_channel(c)[2] for c in range(nch)])
    # ragged blocks
    outs, outq, pos = [], [], 0
    while pos < n:
        m = min(int(rng.randint(1, max_block + 1)), n - pos)
        pcm, q = eng.run(iq[pos:pos + m], m)
        outs.append(pcm)
        outq.append(q)
        pos += m
    eng.close()
    got = np.concatenate(outs, axis=1)
    ref, refq = ora.run_channels(iq, cre, cim, incr, decim, threads=8, want_iq=want_iq)
    desc = "fs %d D %d T %d C %d kernel %s/%d resident %d n %d max_block %d iq %s" % (fs, decim, ntaps, nch, kernel, variant, resident, n,
                                                                                      max_block, want_iq)
    if got.shape != ref.shape or not np.array_equal(got, ref):
        return "PCM differs: " + desc, desc
    qs = [q for q in outq if q is not None]
    if want_iq and refq is not None and refq.shape[1] > 0 and (not qs or not np.array_equal(np.concatenate(qs, axis=1), refq)):
        return "filtered IQ differs: " + desc, desc
    return None, (3 if variant == 1 and resident else 4 if variant == 2 and resident else variant)


def case8(pkg, ora, rng, long_only=False):
    """8-bit ingest: a stream of blocks in the reference's 8-bit formats (mostly one format, now and then another one or
    int16), each block widened for the oracle the way the reference's front ends widen it; geometries drawn so that the
    kernel that reads bytes runs in most cases."""
    b = pkg.binding
    decim = int(rng.choice([32, 64, 96, 96, 96, 128, 160, 25, 40]))
    fs = int(rng.choice([1200000, 2400000]))
    ntaps = max(decim, int(rng.choice([32, 64, 128, 128, 100, 96])))
    if long_only:
        decim = int(rng.choice([4 * int(rng.randint(2, 116)), 8 * int(rng.randint(1, 58)), 96, 400, 25, 100, int(rng.randint(8, 461))]))
        ntaps = max(decim, int(rng.choice([int(rng.randint(257, 513)), 300, 400, 512, 256, 256])))
    nch = int(rng.choice([1, 3, 8, 9, 16, 64, 65, 130]))
    limit = 300000
    if (not long_only) and rng.rand() < 0.12:
        # decimations 1, 2, 4 on 8-bit captures (etc/multifm_file.json is one): the shifted-copies form reading bytes
        decim = int(rng.choice([1, 1, 2, 4]))
        ntaps = int(rng.choice([128, 128, 64, 200, 256, 512]))
        limit = max(4 * ntaps, min(300000, int(2.0e9 * decim / (ntaps * nch))))
    taps = pkg.synth.design_lpf(ntaps, float(rng.choice([5000.0, 12500.0, 40000.0])), fs) * float(rng.choice([1.0, 1.0, 3.0, 0.2]))
    offs = rng.randint(-fs // 2, fs // 2, size=nch)
    offs[: min(nch, 4)] = [0, 25000, -37500, 3125][: min(nch, 4)]
    if fs % (4 * decim) == 0 and rng.rand() < 0.6:
        # channels on the raster of a quarter of the output rate: exact rotators (identity, sign flip, quarter turns), all of
        # them or a random share, so that launches, slices and waves of every class mix occur
        q = fs // (4 * decim)
        snap = rng.rand(nch) < float(rng.choice([1.0, 0.9, 0.5]))
        offs = np.where(snap, q * rng.randint(-2 * decim + 1, 2 * decim, size=nch), offs)
    gains = rng.choice([1.0, 2.5118864315095806, 0.3], size=nch)
    max_block = int(rng.choice([65536, 8192, 100000]))
    flags = (b.MFM_F_WIDEN_8BIT if rng.rand() < 0.15 else 0) | (b.MFM_F_STREAM_TAPS if rng.rand() < 0.2 else 0) | \
            (b.MFM_F_SLICE_128 if rng.rand() < 0.4 else 0)
    try:
        eng = pkg.Engine(fs, decim, max_block, device=0, flags=flags)
        for o, g in zip(offs, gains):
            eng.add_channel(int(o), taps, float(g))
        eng.commit()
    except pkg.MfmError as e:
        return None, "refused: %s" % e
    cre = np.stack([eng.get_channel(c)[0] for c in range(nch)])
    cim = np.stack([eng.get_channel(c)[1] for c in range(nch)])
    incr = np.stack([eng.get_channel(c)[2] for c in range(nch)])
    main_fmt = int(rng.choice([1, 2, 3, 3]))
    total, iq, got, seq = 0, [], [], []
    while total < limit:
        m = int(rng.randint(1, min(max_block, limit) + 1))
        fmt = main_fmt if rng.rand() < 0.9 else int(rng.randint(0, 4))
        if fmt == 0:
            blk = rng.randint(-32768, 32768, size=(m, 2)).astype(np.int16)
            iq.append(blk)
        else:
            blk = rng.randint(0, 256, size=(m, 2)).astype(np.uint8)
            iq.append(ora.unpack_bytes(blk, fmt).reshape(-1, 2))
        seq.append((fmt, m))
        while True:
            rc = eng.push(blk.reshape(-1)) if fmt == 0 else eng.push_bytes(blk, fmt)
            if rc == 0:
                break
            if rc != b.MFM_E_BUSY:
                return "push failed (%d)" % rc, None
            got.append(eng.fetch()[1])
        total += m
    eng.sync()
    while True:
        blk = eng.fetch()
        if blk is None:
            break
        got.append(blk[1])
    st = eng.stats()
    eng.close()
    ref, _ = ora.run_channels(np.concatenate(iq), cre, cim, incr, decim, threads=8)
    gotc = np.concatenate(got, axis=1) if got else np.zeros((nch, 0), np.int16)
    desc = "8-bit: fs %d D %d T %d C %d flags %d max_block %d variant %d bytes-launches %d/%d blocks %s" % (
        fs, decim, ntaps, nch, flags, max_block, st["kernel_variant"], st["launches_8bit"], st["launches"], seq[:12])
    if gotc.shape != ref.shape or not np.array_equal(gotc, ref):
        return "PCM differs: " + desc, desc
    return None, "bytes" if st["launches_8bit"] else "widened"


def case_stream(pkg, ora, rng, long_only=False):
    """The round-4 engine logic: backlog coalescing under both policies, two compute streams, pushes out of page-locked
    memory with copy tickets, flushes at random points, a seek before the stream, front-end sized and ragged blocks in a
    mix of input formats - the PCM stream (and, a third of the time, the filtered IQ) must be that of the oracle on the
    whole input whatever shared a launch."""
    import ctypes
    b = pkg.binding
    lib = pkg.load_library()
    decim = int(rng.choice([96, 96, 96, 64, 128, 40, 25, 100, 32, 8 * int(rng.randint(1, 30)), int(rng.randint(6, 200))]))
    fs = int(rng.choice([1200000, 2400000, 10000000]))
    ntaps = max(decim, int(rng.choice([128, 128, 96, 64, 129, 256, decim + int(rng.randint(0, 60))])))
    nch = int(rng.choice([1, 3, 8, 9, 16, 33, 64, 65, 130, 200]))
    taps = pkg.synth.design_lpf(ntaps, float(rng.choice([5000.0, 12500.0, 40000.0])), fs) * float(rng.choice([1.0, 1.0, 3.0, 0.2]))
    offs = rng.randint(-fs // 2, fs // 2, size=nch)
    offs[: min(nch, 4)] = [0, 25000, -37500, 3125][: min(nch, 4)]
    if fs % (4 * decim) == 0 and rng.rand() < 0.4:
        q = fs // (4 * decim)
        snap = rng.rand(nch) < float(rng.choice([1.0, 0.9, 0.5]))
        offs = np.where(snap, q * rng.randint(-2 * decim + 1, 2 * decim, size=nch), offs)
    gains = rng.choice([1.0, 2.5118864315095806, 0.3], size=nch)
    want_iq = bool(rng.rand() < 0.33)
    max_block = int(rng.choice([4096, 16384, 131072, 262144, 100000]))
    coalesce = int(rng.choice([0, max_block, 4 * max_block, 300000, 1 << 20, int(rng.randint(1, 500000))]))
    flags = int(rng.choice([0, 0, b.MFM_F_FORCE_MFMA_V1, b.MFM_F_FORCE_DOT2]))
    flags |= b.MFM_F_GATHER if rng.rand() < 0.5 else 0
    flags |= b.MFM_F_OVERLAP if rng.rand() < 0.5 else 0
    flags |= b.MFM_F_WIDEN_8BIT if rng.rand() < 0.1 else 0
    flags |= b.MFM_F_STREAM_TAPS if rng.rand() < 0.1 else 0
    flags |= b.MFM_F_SLICE_128 if rng.rand() < 0.35 else 0
    try:
        eng = pkg.Engine(fs, decim, max_block, device=0, flags=flags, coalesce_samples=coalesce)
        for o, g in zip(offs, gains):
            eng.add_channel(int(o), taps, float(g), want_iq=want_iq)
        eng.commit()
    except pkg.MfmError as e:
        return None, "refused: %s" % e
    chans = [eng.get_channel(c) for c in range(nch)]
    before = 0
    if rng.rand() < 0.3:
        # a stream to forget, then a resume somewhere: inside the pre-period, around 2^32, far out
        eng.run(pkg.synth.random_iq(3 * ntaps + 1000, seed=3), min(max_block, 1000))
        before = int(rng.choice([0, 1, int(rng.randint(1, 100000)), (1 << 32) - int(rng.randint(0, 50)), int(rng.randint(1 << 33, 1 << 45))]))
        eng.seek(before)
    main_fmt = int(rng.choice([0, 0, 1, 2, 3]))
    pinned = bool(rng.rand() < 0.5)
    npool = 4
    pool = [lib.mfm_host_alloc(max_block * 4) for _ in range(npool)] if pinned else []
    tickets = [0] * npool
    total_want = int(rng.randint(ntaps, 300000 if before else 1500000))  # the resumed oracle runs channel by channel
    sizes_menu = [1, 2, 4096, 4096, 16384, 131072, 262144, int(rng.randint(1, max_block + 1)), ntaps - 1, ntaps, decim, max_block]
    total, iq, parts, qparts, k = 0, [], [], [], 0

    def drain():
        while True:
            got = eng.fetch()
            if got is None:
                return None
            if parts and got[0] != parts[-1][0] + parts[-1][1].shape[1]:
                return "block starts at output %d, the previous one ended at %d" % (got[0], parts[-1][0] + parts[-1][1].shape[1])
            parts.append((got[0], got[1]))
            if got[2] is not None:
                qparts.append(got[2])

    err = None
    while total < total_want and err is None:
        m = max(1, min(int(rng.choice(sizes_menu)), max_block, total_want - total))
        fmt = main_fmt if rng.rand() < 0.93 else int(rng.randint(0, 4))
        if fmt == 0:
            blk = rng.randint(-32768, 32768, size=(m, 2)).astype(np.int16)
            iq.append(blk)
        else:
            blk = rng.randint(0, 256, size=(m, 2)).astype(np.uint8)
            iq.append(ora.unpack_bytes(blk, fmt).reshape(-1, 2))
        slot = k % npool
        if pinned:
            if lib.mfm_engine_copy_wait(eng.h, tickets[slot]) != 0:
                err = "copy_wait failed"
                break
            ctypes.memset(pool[slot], 0xa5, max_block * 4)
            ctypes.memmove(pool[slot], blk.ctypes.data, blk.nbytes)
        while True:
            if pinned:
                t = ctypes.c_uint64()
                rc = lib.mfm_engine_push_pinned(eng.h, pool[slot], m, fmt, ctypes.byref(t))
                if rc == 0:
                    tickets[slot] = t.value
            else:
                rc = eng.push(blk.reshape(-1)) if fmt == 0 else eng.push_bytes(blk, fmt)
            if rc == 0:
                break
            if rc != b.MFM_E_BUSY:
                err = "push failed (%d): %s" % (rc, lib.mfm_last_error())
                break
            err = drain()
            if err:
                break
        total += m
        k += 1
        if rng.rand() < 0.03:
            eng.flush()
        if rng.rand() < 0.2:
            err = err or drain()
    while err is None:
        rc = eng.flush()
        err = drain()
        if rc == 0:
            break
    eng.sync()
    err = err or drain()
    st = eng.stats()
    eng.close()
    for p in pool:
        lib.mfm_host_free(p)
    desc = "stream: fs %d D %d T %d C %d flags 0x%x max_block %d coalesce %d pinned %d fmt %d seek %d n %d variant %d launches %d/%d submits iq %s" % (
        fs, decim, ntaps, nch, flags, max_block, coalesce, pinned, main_fmt, before, total, st["kernel_variant"], st["launches"],
        st["submits"], want_iq)
    if err:
        return err + ": " + desc, desc
    x = np.concatenate(iq)
    pcm = np.concatenate([p[1] for p in parts], axis=1) if parts else np.zeros((nch, 0), np.int16)
    if parts and parts[0][0] != before:
        return "first block at output %d: %s" % (parts[0][0], desc), desc
    if 0 == before:
        cre, cim, incr = (np.stack([c[i] for c in chans]) for i in range(3))
        ref, refq = ora.run_channels(x, cre, cim, incr, decim, threads=8, want_iq=want_iq)
    else:
        refs, refqs = [], []
        for cre, cim, incr in chans:
            ch = ora.Channel(cre, cim, decim, incr)
            ch.skip_outputs(before)
            w, wq = ch.feed(x)
            ch.close()
            refs.append(w)
            refqs.append(wq.reshape(-1, 2))
        ref, refq = np.stack(refs), (np.stack(refqs) if want_iq else None)
    if pcm.shape != ref.shape or not np.array_equal(pcm, ref):
        return "PCM differs: " + desc, desc
    if want_iq and refq is not None and refq.shape[1] > 0:
        gq = np.concatenate(qparts, axis=1) if qparts else None
        if gq is None or not np.array_equal(gq.reshape(refq.shape), refq):
            return "filtered IQ differs: " + desc, desc
    return None, "v%d%s%s%s" % (st["kernel_variant"], "+gather" if flags & b.MFM_F_GATHER else "", "+overlap" if flags & b.MFM_F_OVERLAP else "",
                                "+pinned" if pinned else "")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=300.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--ingest8", action="store_true", help="8-bit ingest streams (mfm_engine_push_bytes) instead")
    ap.add_argument("--stream", action="store_true", help="coalescing / two streams / pinned pushes / flush / seek (round 4 engine logic)")
    ap.add_argument("--long", action="store_true", help="filters of 129..512 taps only (resident and streamed tap instances)")
    ap.add_argument("--slice128", action="store_true", help="filters of up to 128 taps on 128-channel slices (round 6: MFM_F_SLICE_128)")
    args = ap.parse_args()
    from __graft_entry__ import load_package
    import oracle_lib as ora
    pkg = load_package()
    rng = np.random.RandomState(args.seed)
    t0 = time.time()
    counts = {}
    while time.time() - t0 < args.seconds:
        if args.stream:
            err, info = case_stream(pkg, ora, rng, args.long)
        else:
            err, info = case8(pkg, ora, rng, args.long) if args.ingest8 else case(pkg, ora, rng, args.long, args.slice128)
        if err:
            print("FAIL", err, "after", counts)
            return 1
        if args.ingest8 or args.stream:
            k8 = "refused" if str(info).startswith("refused") else info
            counts[k8] = counts.get(k8, 0) + 1
            continue
        key = info if isinstance(info, (int, np.integer)) else "refused"
        counts[int(key) if key != "refused" else key] = counts.get(int(key) if key != "refused" else key, 0) + 1
    print("ok: cases per kernel variant (0 v_dot2, 1 matrix gen 1, 2 matrix gen 2, 3 matrix gen 1 with resident taps, 4 matrix gen 2 long-filter kernel; --ingest8: by how the blocks were read)", counts)
    return 0


if __name__ == "__main__":
    sys.exit(main())
