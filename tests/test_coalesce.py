"""Backlog coalescing (mfm_engine_config::coalesce_samples): blocks accepted into the buffer being filled and launched as
one pass.  The reference's channel thread takes whatever its work queue holds - up to 128 sample_bufs of 4096 (file_if.c:18),
131072 (rtl_sdr_if.c:46) or 262144 (airspy_if.c:244-245) samples - and runs them back to back (multifm/demod.c:48-121,
134-150, :297): its output does not depend on the buffers' sizes, and neither may the engine's depend on which blocks
shared a launch.  Bit-exact against the oracle, through the C ABI."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tables(ora, taps, offs, fs, decim, gains=None):
    gains = gains if gains is not None else [1.0] * len(offs)
    cre = np.stack([ora.make_taps(taps, int(o), fs, float(g))[0] for o, g in zip(offs, gains)])
    cim = np.stack([ora.make_taps(taps, int(o), fs, float(g))[1] for o, g in zip(offs, gains)])
    incr = np.stack([ora.rot_incr(int(o), fs, decim) for o in offs])
    return cre, cim, incr


def _drain(eng, parts, qparts):
    while True:
        got = eng.fetch()
        if got is None:
            return
        if parts and got[0] != parts[-1][0] + parts[-1][1].shape[1]:
            raise AssertionError(f"block starts at output {got[0]}, the previous one ended at "
                                 f"{parts[-1][0] + parts[-1][1].shape[1]}")
        parts.append((got[0], got[1]))
        if got[2] is not None:
            qparts.append(got[2])


def _finish(eng, parts, qparts):
    while True:
        rc = eng.flush()
        _drain(eng, parts, qparts)
        if rc == 0:
            break
    eng.sync()
    _drain(eng, parts, qparts)


SIZES = [1, 1, 100, 26, 1, 4096, 4096, 4096, 131072, 16384, 7919, 5, 127, 128, 129, 60000, 1000, 4096, 4096, 262144]


@pytest.mark.parametrize("kernel", ["auto", "mfma1", "dot2"])
@pytest.mark.parametrize("policy", ["gather", "auto", "gather+overlap", "auto+overlap"])
def test_coalesced_ragged_buffers_against_the_oracle(pkg, ora, kernel, policy):
    """Front-end sized buffers (file_if 4096, rtl_sdr 131072, uhd 16384, airspy 262144), single samples and blocks shorter
    than the filter, gathered into launches of up to 300 000 samples: same PCM and filtered IQ as the oracle on the whole
    stream; with MFM_F_GATHER the launch boundaries are those of the gathered sample count, not of the blocks."""
    b = pkg.binding
    fs, decim, taps, offs, gains = pkg.synth.plan("cfg2_64ch", nr_channels=9)
    n = 1200000
    iq = pkg.synth.synth_iq(n, fs, offs[:3], seed=41)
    flags = {"auto": 0, "mfma1": b.MFM_F_FORCE_MFMA_V1, "dot2": b.MFM_F_FORCE_DOT2}[kernel]
    flags |= b.MFM_F_GATHER if policy.startswith("gather") else 0
    flags |= b.MFM_F_OVERLAP if policy.endswith("overlap") else 0   # two compute streams (second-generation kernel only)
    eng = pkg.Engine(fs, decim, 262144, device=0, flags=flags, coalesce_samples=300000)
    for o, g in zip(offs, gains):
        eng.add_channel(int(o), taps, float(g), want_iq=True)
    eng.commit()
    cre = np.stack([eng.get_channel(c)[0] for c in range(len(offs))])
    cim = np.stack([eng.get_channel(c)[1] for c in range(len(offs))])
    incr = np.stack([eng.get_channel(c)[2] for c in range(len(offs))])
    ref, refq = ora.run_channels(iq, cre, cim, incr, decim, threads=4, want_iq=True)
    parts, qparts, pos, k = [], [], 0, 0
    while pos < n:
        m = min(SIZES[k % len(SIZES)], n - pos)
        rc = eng.push(iq[pos:pos + m])
        if rc == b.MFM_E_BUSY:
            _drain(eng, parts, qparts)
            continue
        assert rc == 0, eng.lib.mfm_last_error()
        pos += m
        k += 1
    _finish(eng, parts, qparts)
    st = eng.stats()
    eng.close()
    pcm = np.concatenate([p[1] for p in parts], axis=1)
    q = np.concatenate(qparts, axis=1)
    assert parts[0][0] == 0 and pcm.shape == ref.shape
    assert np.array_equal(pcm, ref) and np.array_equal(q, refq)
    assert st["samples_in"] == n and st["outputs"] == ref.shape[1] and st["submits"] == k and st["pending_samples"] == 0
    assert st["launches"] <= st["submits"]
    if policy.startswith("gather"):
        # a launch per 300 000 gathered samples (plus the block that crossed the mark), and the flush at the end
        assert st["launches"] <= n // 300000 + 1, st["launches"]
        assert all(p[1].shape[1] >= 300000 // decim - 2 for p in parts[:-1])


@pytest.mark.parametrize("policy", ["gather", "auto"])
def test_coalesced_8bit_blocks_and_format_changes(pkg, ora, policy):
    """8-bit blocks read as bytes by the matrix kernel gather like int16 ones; a block of another format behind accepted
    ones (int16 behind RTL-SDR bytes, cs8 behind cu8, a cu8 block of odd length) sends what was gathered out as a launch
    of its own and the stream goes on - the oracle on the reference's host-side widening sees one stream."""
    b = pkg.binding
    fs, decim = 2400000, 96
    taps = pkg.synth.design_lpf(128, 9000.0, fs)
    offs = [-300000, 12500, 412500]
    rng = np.random.RandomState(78)

    def raw(m):
        return rng.randint(0, 256, size=(m, 2)).astype(np.uint8)

    def s16(m):
        return rng.randint(-32768, 32768, size=(m, 2)).astype(np.int16)

    blocks = [(raw(20000), 3), (raw(20002), 3), (raw(4096), 3), (s16(7777), 0), (s16(4096), 0), (raw(9000), 3), (raw(4096), 1),
              (raw(4098), 1), (raw(4097), 2), (raw(4098), 2), (raw(10), 1), (s16(96 * 50), 0), (raw(30000), 1), (raw(30000), 1)]
    eng = pkg.Engine(fs, decim, 32768, device=0, flags=b.MFM_F_GATHER if policy == "gather" else 0, coalesce_samples=100000)
    for o in offs:
        eng.add_channel(int(o), taps, 1.0)
    eng.commit()
    iq, parts, qparts = [], [], []
    for blk, fmt in blocks:
        iq.append(blk.astype(np.int16).reshape(-1, 2) if fmt == 0 else ora.unpack_bytes(blk, fmt).reshape(-1, 2))
        while True:
            rc = eng.push(blk.reshape(-1)) if fmt == 0 else eng.push_bytes(blk, fmt)
            if rc == 0:
                break
            assert rc == b.MFM_E_BUSY
            _drain(eng, parts, qparts)
    _finish(eng, parts, qparts)
    st = eng.stats()
    eng.close()
    iq = np.concatenate(iq)
    cre, cim, incr = _tables(ora, taps, offs, fs, decim)
    want, _ = ora.run_channels(iq, cre, cim, incr, decim)
    got = np.concatenate([p[1] for p in parts], axis=1)
    assert got.shape == want.shape and np.array_equal(got, want)
    assert st["submits"] == len(blocks) and st["launches"] <= st["submits"]
    if policy == "gather":
        # the three RTL-SDR blocks in front share ONE byte launch; behind an int16 history everything is widened
        assert 1 == st["launches_8bit"] < st["launches"] < st["submits"]


@pytest.mark.parametrize("overlap", [False, True])
@pytest.mark.parametrize("nch,block_log2,coalesce_log2", [(64, 12, 20), (16, 17, 22), (130, 14, 18), (64, 20, 0)])
def test_device_resident_blocks_replayed_from_c(pkg, ora, nch, block_log2, coalesce_log2, overlap):
    """mfm_engine_replay: the producer loop of a C host - acquire_input, submit - on blocks that are already in HBM (what
    bench.py's block_series times), with the engine's own launch policy.  Every launch's PCM in HBM against the oracle on
    the samples that launch read (mfm_engine_last_launch_input), rotators stepped to the launch's first output."""
    b = pkg.binding
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    fs, decim, taps, offs, gains = pkg.synth.plan("cfg2_64ch", nr_channels=nch)
    block, co = 1 << block_log2, (1 << coalesce_log2) if coalesce_log2 else 0
    eng = pkg.Engine(fs, decim, block, device=0, flags=b.MFM_F_DEVICE_ONLY | (b.MFM_F_OVERLAP if overlap else 0),
                     coalesce_samples=co)
    for o, g in zip(offs, gains):
        eng.add_channel(int(o), taps, float(g))
    eng.commit()
    cfg = b.EngineConfig()
    cfg.max_block_samples, cfg.coalesce_samples = block, co
    nb = C.c_uint32()
    in_bytes = eng.lib.mfm_engine_input_bytes_cfg(C.byref(cfg), len(taps), C.byref(nb))
    assert nb.value == (3 if co else 2) and in_bytes >= 4 * (block + co + len(taps))
    # every input buffer holds the same synthetic samples (the buffers are the engine's own: fill them where it says)
    base = pkg.synth.synth_iq(in_bytes // 4, fs, offs[:4], seed=5).reshape(-1)
    seen = set()
    for _ in range(3):
        ptr, cap = eng.acquire_input()
        assert cap == block
        if ptr not in seen:
            seen.add(ptr)
            assert hip.hipMemcpy(ptr, base.ctypes.data, in_bytes - 4 * (2 * len(taps) + 64), 1) == 0
        eng.submit(block, producer_stream=0, wait_producer=False)
        eng.flush()
    eng.sync()
    eng.reset()
    checked = 0
    for rounds, blocks in ((1, 1), (1, 7), (3, 150), (2, 1000 if block_log2 < 20 else 40)):
        for _ in range(rounds):
            eng.replay(block, blocks)
        eng.sync()
        st = eng.stats()
        dptr, stride, nout, _ = eng.last_output_device()
        iptr, n_in, fmt = eng.last_launch_input()
        assert fmt == 0 and nout == (n_in - len(taps)) // decim + 1
        host_in = np.empty((n_in, 2), np.int16)
        assert hip.hipMemcpy(host_in.ctypes.data, iptr, 4 * n_in, 2) == 0
        before = st["outputs"] - nout
        for c in sorted(set([0, nch // 2, nch - 1])):
            row = np.empty(nout, np.int16)
            assert hip.hipMemcpy(row.ctypes.data, C.c_void_p(dptr + 2 * stride * c), 2 * nout, 2) == 0
            cre, cim, incr = eng.get_channel(c)
            for w0 in sorted(set([1, max(1, nout // 2), max(1, nout - 700)])):
                cnt = min(700, nout - w0)
                want = ora.window_pcm(host_in, cre, cim, decim, incr, before, w0, cnt)
                assert np.array_equal(row[w0:w0 + cnt], want), (c, w0, blocks)
                checked += cnt
    st = eng.stats()
    eng.close()
    assert checked > 0 and st["submits"] >= 3 * 150 + 2 * 40 and st["pending_samples"] == 0
    if co:
        assert st["launches"] < st["submits"]  # 2000 blocks handed over faster than the device takes single ones
    else:
        assert st["launches"] == st["submits"]


@pytest.mark.parametrize("shards,mode,nch,gather", [(2, "rccl", 70, "gather"), (3, "allgather", 70, "auto"), (8, "allgather", 130, "gather")])
def test_coalescing_group_of_several_shards_through_a_fake_transport(tmp_path, shards, mode, nch, gather):
    """The device group with coalesce_samples over the test double of RCCL (tests/test_group.py): every shard accepts and
    launches the same blocks together (one decision per block for all shards, mfm_group_seq.h), int16 and 8-bit blocks,
    format changes flushed on every shard at once; concatenated PCM against the oracle."""
    so = tmp_path / "librccl.so"
    r = subprocess.run(["/opt/rocm/bin/hipcc", "-O2", "-fPIC", "-shared", "-o", str(so),
                        os.path.join(ROOT, "tests", "hoststub", "fake_rccl.cpp")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    env = dict(os.environ, LD_LIBRARY_PATH=str(tmp_path) + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run(["python3", os.path.join(ROOT, "tests", "hoststub", "multi_shard_run.py"), str(shards), mode, str(nch),
                        "150000", gather], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "multi-shard ok" in r.stdout, (r.stdout + r.stderr)[-3000:]


@pytest.mark.parametrize("want_iq", [False, True])
@pytest.mark.parametrize("plan,nch", [("cfg2_64ch", 64), ("cfg2_64ch_grid", 64), ("cfg3_1024ch", 200), ("multifm_1ch", 20)])
def test_overlapped_launches_against_the_oracle(pkg, ora, plan, nch, want_iq):
    """MFM_F_OVERLAP: consecutive launches on two streams.  A launch takes nothing from the one before it but input samples
    (the output in front of it is recomputed from the row kept in front of the first unconsumed sample, the rotator position
    folded from the stream's output count), so whatever order the device runs their workgroups in, the stream is the
    oracle's: general, quarter-turn, sign-flip and identity rotators (filter/direct_fir.c:151-172), filtered IQ on and off
    (different kernel instances), blocks of 20 tiles and of less than one, 16 slices of channels."""
    b = pkg.binding
    fs, decim, taps, offs, gains = pkg.synth.plan(plan, nr_channels=nch)
    n = decim * 40000 + 777
    iq = pkg.synth.synth_iq(n, fs, offs[:3], seed=43)
    eng = pkg.Engine(fs, decim, 1 << 17, device=0, flags=b.MFM_F_OVERLAP)
    for o, g in zip(offs, gains):
        eng.add_channel(int(o), taps, float(g), want_iq=want_iq)
    eng.commit()
    assert eng.stats()["kernel_variant"] == 2
    cre = np.stack([eng.get_channel(c)[0] for c in range(nch)])
    cim = np.stack([eng.get_channel(c)[1] for c in range(nch)])
    incr = np.stack([eng.get_channel(c)[2] for c in range(nch)])
    ref, refq = ora.run_channels(iq, cre, cim, incr, decim, threads=8, want_iq=want_iq)
    sizes = [131072, 131072, 5000, 131072, 64 * decim, 63 * decim, 1, 131072, 100, 131072, 131072, 90000]
    parts, qparts, pos, k = [], [], 0, 0
    while pos < n:
        m = min(sizes[k % len(sizes)], n - pos)
        rc = eng.push(iq[pos:pos + m])
        if rc == b.MFM_E_BUSY:
            _drain(eng, parts, qparts)
            continue
        assert rc == 0, eng.lib.mfm_last_error()
        pos += m
        k += 1
    _finish(eng, parts, qparts)
    eng.close()
    pcm = np.concatenate([p[1] for p in parts], axis=1)
    assert pcm.shape == ref.shape
    if not np.array_equal(pcm, ref):
        bad = np.argwhere(pcm != ref)
        raise AssertionError(f"{len(bad)} PCM samples differ; first at (chan, n) = {bad[0]}")
    if want_iq:
        assert np.array_equal(np.concatenate(qparts, axis=1), refq)


@pytest.mark.parametrize("kernel", ["auto", "mfma1", "dot2", "auto+overlap"])
@pytest.mark.parametrize("before", [0, 53000, (1 << 32) - 300, (1 << 33) + 12345, (1 << 40) + 1])
def test_seek_resumes_the_rotators_also_beyond_2_to_the_32(pkg, ora, kernel, before):
    """mfm_engine_seek: a stream resumed at output `before` - the derotators where that many steps of the recurrence leave
    them (filter/direct_fir.c:166-167), history and discriminator empty.  Counts inside the pre-period of the 101 kHz
    rotator (53 105 steps), across 2^32 (the kernels fold a 64-bit output count; a stream at the bench's rate gets there
    within a second) and far beyond.  Oracle: the same channel stepped to `before` (mfmo_chan_skip_outputs)."""
    b = pkg.binding
    fs, decim = 2400000, 96
    taps = pkg.synth.design_lpf(128, 12500.0, fs)
    offs = [101000, 777, 3125, 37500, 25000, 12500, -6250, 1000000]
    n = decim * 900 + 500
    iq = pkg.synth.synth_iq(n, fs, offs[:2], seed=52, noise=2000)
    flags = {"auto": 0, "mfma1": b.MFM_F_FORCE_MFMA_V1, "dot2": b.MFM_F_FORCE_DOT2, "auto+overlap": b.MFM_F_OVERLAP}[kernel]
    eng = pkg.Engine(fs, decim, 1 << 15, device=0, flags=flags)
    for o in offs:
        eng.add_channel(int(o), taps, 1.0, want_iq=True)
    eng.commit()
    # a stream first, so that seek has something to forget
    eng.run(iq[:20000], 7000)
    eng.seek(before)
    parts, qparts = [], []
    for lo in range(0, n, 30000):
        assert eng.push(iq[lo:lo + 30000]) == 0
        _drain(eng, parts, qparts)
    _finish(eng, parts, qparts)
    assert parts[0][0] == before
    pcm = np.concatenate([p[1] for p in parts], axis=1)
    q = np.concatenate(qparts, axis=1)
    for c in range(len(offs)):
        cre, cim, incr = eng.get_channel(c)
        ch = ora.Channel(cre, cim, decim, incr)
        ch.skip_outputs(before)
        want, wantq = ch.feed(iq)
        ch.close()
        assert np.array_equal(pcm[c], want), (c, before)
        assert np.array_equal(q[c], wantq.reshape(-1, 2)), (c, before)
    eng.close()


@pytest.mark.parametrize("coalesce", [0, 200000])
def test_pushes_out_of_page_locked_memory_with_copy_tickets(pkg, ora, coalesce):
    """mfm_engine_push_pinned: the H2D copy reads the caller's page-locked buffer in place (what the C host's sample_buf pool
    is) and a ticket says when the buffer may be reused - the front end's next fill, here an overwrite with garbage the
    moment the ticket is through.  int16 and RTL-SDR byte blocks; PCM against the oracle."""
    import ctypes
    b = pkg.binding
    lib = pkg.load_library()
    fs, decim, taps, offs, gains = pkg.synth.plan("cfg2_64ch", nr_channels=12)
    nbuf, spb = 4, 16384
    pool = [lib.mfm_host_alloc(spb * 4) for _ in range(nbuf)]
    assert all(pool)
    eng = pkg.Engine(fs, decim, spb, device=0, coalesce_samples=coalesce)
    for o, g in zip(offs, gains):
        eng.add_channel(int(o), taps, float(g))
    eng.commit()
    rng = np.random.RandomState(5)
    sizes = [spb, spb, 4096, spb, 100, spb, 7777, spb] * 6
    iq, parts, qparts, tickets = [], [], [], [0] * nbuf
    for k, m in enumerate(sizes):
        fmt = b.MFM_IN_RTLSDR_U8 if (k // 8) % 2 else b.MFM_IN_CS16
        slot = k % nbuf
        assert lib.mfm_engine_copy_wait(eng.h, tickets[slot]) == 0 and lib.mfm_engine_copy_done(eng.h, tickets[slot]) == 1
        if fmt == b.MFM_IN_CS16:
            blk = rng.randint(-20000, 20000, size=(m, 2)).astype(np.int16)
            iq.append(blk)
        else:
            blk = rng.randint(0, 256, size=(m, 2)).astype(np.uint8)
            iq.append(ora.unpack_bytes(blk, fmt).reshape(-1, 2))
        ctypes.memset(pool[slot], 0x5a, spb * 4)   # whatever was there is gone: its copy had better be through
        ctypes.memmove(pool[slot], blk.ctypes.data, blk.nbytes)
        t = ctypes.c_uint64()
        while True:
            rc = lib.mfm_engine_push_pinned(eng.h, pool[slot], m, fmt, ctypes.byref(t))
            if rc == 0:
                break
            assert rc == b.MFM_E_BUSY, lib.mfm_last_error()
            _drain(eng, parts, qparts)
        tickets[slot] = t.value
        assert t.value == k + 1
    _finish(eng, parts, qparts)
    eng.close()
    for p in pool:
        lib.mfm_host_free(p)
    iq = np.concatenate(iq)
    cre, cim, incr = _tables(ora, taps, offs, fs, decim, gains)
    want, _ = ora.run_channels(iq, cre, cim, incr, decim, threads=4)
    got = np.concatenate([p[1] for p in parts], axis=1)
    assert got.shape == want.shape and np.array_equal(got, want)


@pytest.mark.parametrize("coalesce,max_run", [(0, 4), (300000, 16), (300000, 64)])
def test_runs_of_neighbouring_pool_frames_as_one_copy_command(pkg, ora, coalesce, max_run):
    """mfm_engine_push_pinned_run (round 5): frames of ONE page-locked arena that lie `stride` bytes apart - a sample_buf header
    between the data of two frames, as in host/mfm_receiver.c's pool - go to the device as one strided copy command and are
    accepted as one block.  `accepted` may be smaller than offered (the buffer being filled is full, or the gathering policy
    wants a launch in between): the rest is offered again.  int16 and byte frames, frames reused (overwritten) as soon as their
    ticket is through; the PCM stream against the oracle."""
    import ctypes
    b = pkg.binding
    lib = pkg.load_library()
    fs, decim, taps, offs, gains = pkg.synth.plan("cfg2_64ch", nr_channels=10)
    nfr, spf, hdr = 96, 4096, 48                       # frames of 4096 samples with a 48-byte header in front of each
    stride = hdr + spf * 4
    arena = lib.mfm_host_alloc(nfr * stride)
    assert arena
    eng = pkg.Engine(fs, decim, spf * max_run, device=0, coalesce_samples=coalesce)
    for o, g in zip(offs, gains):
        eng.add_channel(int(o), taps, float(g))
    eng.commit()
    rng = np.random.RandomState(11)
    iq, parts, qparts = [], [], []
    tickets = [0] * nfr
    commands = offered = 0
    for rnd in range(3):
        fmt = b.MFM_IN_CS16 if rnd != 1 else b.MFM_IN_RTLSDR_U8
        per = spf if fmt == b.MFM_IN_CS16 else 2 * spf     # a frame of bytes holds twice the samples
        # the front end fills every frame of the pool (after its last copy is through) ...
        for f in range(nfr):
            assert lib.mfm_engine_copy_wait(eng.h, tickets[f]) == 0
            if fmt == b.MFM_IN_CS16:
                blk = rng.randint(-20000, 20000, size=(per, 2)).astype(np.int16)
                iq.append(blk)
            else:
                blk = rng.randint(0, 256, size=(per, 2)).astype(np.uint8)
                iq.append(ora.unpack_bytes(blk, fmt).reshape(-1, 2))
            ctypes.memset(arena + f * stride, 0xa5, stride)
            ctypes.memmove(arena + f * stride + hdr, blk.ctypes.data, blk.nbytes)
        # ... and the submit thread sends them in runs
        f = 0
        while f < nfr:
            want_n = min(int(rng.randint(1, max_run + 1)), nfr - f)
            t, took = ctypes.c_uint64(), ctypes.c_size_t()
            rc = lib.mfm_engine_push_pinned_run(eng.h, arena + f * stride + hdr, stride, per, want_n, fmt, ctypes.byref(t), ctypes.byref(took))
            if rc == b.MFM_E_BUSY:
                _drain(eng, parts, qparts)
                continue
            assert rc == 0, lib.mfm_last_error()
            assert 1 <= took.value <= want_n
            for k in range(took.value):
                tickets[f + k] = t.value
            f += took.value
            commands += 1
            offered += want_n
    _finish(eng, parts, qparts)
    if coalesce:
        assert commands < 3 * nfr                      # fewer commands than frames: runs were taken
    else:
        assert commands == 3 * nfr                     # without gathering every frame is a launch of its own: one is accepted at a time
    eng.close()
    lib.mfm_host_free(arena)
    iq = np.concatenate(iq)
    cre, cim, incr = _tables(ora, taps, offs, fs, decim, gains)
    want, _ = ora.run_channels(iq, cre, cim, incr, decim, threads=4)
    got = np.concatenate([p[1] for p in parts], axis=1)
    assert got.shape == want.shape and np.array_equal(got, want)


@pytest.mark.parametrize("geoms,kernel", [(((48, 64), (56, 64)), "auto"), (((25, 128), (30, 128), (12, 40)), "mfma1"),
                                          (((96, 128), (96, 100), (32, 32)), "auto")])
def test_two_engines_sharing_a_kernel_instance_keep_their_lds(pkg, ora, geoms, kernel):
    """the dynamic-LDS limit belongs to a kernel instance on a device, not to an engine.  Engines whose
    geometries select the same template instance with different LDS sizes (the generic second-generation instances for
    decimations 48 and 56; first-generation instances at different decimations) live side by side, the one that needs less
    committed LAST - the limit is only ever raised, so the first engine's launches still fit; all of them against the oracle,
    interleaved."""
    b = pkg.binding
    fs = 1200000
    flags = b.MFM_F_FORCE_MFMA_V1 if kernel == "mfma1" else 0
    engs = []
    for decim, ntaps in sorted(geoms, key=lambda g: -g[0]):      # the largest image first, the smallest last
        taps = pkg.synth.design_lpf(ntaps, 9000.0, fs)
        offs = [12345, -250000, 100000, 37500, 0]
        eng = pkg.Engine(fs, decim, 1 << 15, device=0, flags=flags)
        for o in offs:
            eng.add_channel(int(o), taps, 1.0)
        eng.commit()
        iq = pkg.synth.synth_iq(decim * 700 + ntaps + 5, fs, offs[:3], seed=decim)
        engs.append((eng, decim, taps, offs, iq))
    for rnd in range(2):
        for eng, decim, taps, offs, iq in engs:
            eng.reset()
            pcm, _ = eng.run(iq, 9000)
            cre = np.stack([eng.get_channel(c)[0] for c in range(len(offs))])
            cim = np.stack([eng.get_channel(c)[1] for c in range(len(offs))])
            incr = np.stack([eng.get_channel(c)[2] for c in range(len(offs))])
            ref, _ = ora.run_channels(iq, cre, cim, incr, decim)
            assert pcm.shape == ref.shape and np.array_equal(pcm, ref), (decim, rnd)
    for eng, *_ in engs:
        eng.close()


def test_engines_driven_from_their_own_threads(pkg, ora):
    """Several receivers in one process (one engine each, as one multifm process per dongle would be folded into one service):
    every engine is created, committed, fed and drained from its own thread while the others do the same - creation and commit
    included, so the process-wide pieces (kernel selection, the dynamic-LDS limits, the division self-test, the last-error
    slot) are exercised concurrently.  Each engine's PCM against the oracle."""
    import threading
    b = pkg.binding
    shapes = [(2400000, 96, 128, 33, 0), (1000000, 40, 128, 9, 0), (1200000, 25, 128, 5, b.MFM_F_OVERLAP),
              (1200000, 25, 256, 3, 0), (2500000, 100, 256, 4, 0), (2400000, 96, 128, 8, b.MFM_F_FORCE_DOT2),
              (2400000, 96, 128, 70, b.MFM_F_GATHER), (1000000, 40, 64, 2, b.MFM_F_FORCE_MFMA_V1)]
    results, errors = {}, []

    def work(k, fs, decim, ntaps, nch, flags):
        try:
            rng = np.random.RandomState(k)
            taps = pkg.synth.design_lpf(ntaps, 9000.0, fs)
            offs = rng.randint(-fs // 2 + 30000, fs // 2 - 30000, size=nch)
            iq = pkg.synth.random_iq(decim * 3000 + ntaps + k, seed=100 + k)
            eng = pkg.Engine(fs, decim, 1 << 16, device=0, flags=flags, coalesce_samples=100000 if flags & b.MFM_F_GATHER else 0)
            for o in offs:
                eng.add_channel(int(o), taps, 1.0)
            eng.commit()
            tabs = [eng.get_channel(c) for c in range(nch)]
            parts, qparts, pos = [], [], 0
            while pos < iq.shape[0]:
                m = min(int(rng.randint(1, 1 << 16)), iq.shape[0] - pos)
                rc = eng.push(iq[pos:pos + m])
                if rc == b.MFM_E_BUSY:
                    _drain(eng, parts, qparts)
                    continue
                assert rc == 0, eng.lib.mfm_last_error()
                pos += m
            _finish(eng, parts, qparts)
            eng.close()
            results[k] = (np.concatenate([p[1] for p in parts], axis=1), iq, tabs, decim)
        except Exception as e:  # noqa: BLE001 - reported by the main thread
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=work, args=(k,) + s) for k, s in enumerate(shapes)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(300)
    assert not errors, errors
    assert len(results) == len(shapes)
    for k, (pcm, iq, tabs, decim) in results.items():
        cre, cim, incr = (np.stack([t[i] for t in tabs]) for i in range(3))
        ref, _ = ora.run_channels(iq, cre, cim, incr, decim, threads=4)
        assert pcm.shape == ref.shape and np.array_equal(pcm, ref), shapes[k]


def test_link_probe_and_input_room(pkg):
    """mfm_link_probe / mfm_link_probe_runs (what bench.py's end_to_end.link reports): positive rates for the command shapes the
    host uses; mfm_engine_input_room: what the buffer being filled still takes goes down by what was pushed and comes back
    with the launch."""
    import ctypes
    lib = pkg.load_library()
    h2d, d2h = ctypes.c_double(), ctypes.c_double()
    assert lib.mfm_link_probe(0, 512 << 10, 64 << 20, 1.0 / 3.0, ctypes.byref(h2d), ctypes.byref(d2h)) == 0, lib.mfm_last_error()
    assert 1.0 < h2d.value < 200.0 and 0.3 < d2h.value < 200.0, (h2d.value, d2h.value)
    one = h2d.value
    assert lib.mfm_link_probe_runs(0, 512 << 10, 48, 16, 64 << 20, 1.0 / 3.0, ctypes.byref(h2d), ctypes.byref(d2h)) == 0, lib.mfm_last_error()
    assert h2d.value > 0.8 * one, (one, h2d.value)        # runs of 16 frames per command are not slower than one frame per command
    assert lib.mfm_link_probe(0, 0, 1 << 20, 0.0, ctypes.byref(h2d), ctypes.byref(d2h)) != 0
    fs, decim, taps, offs, gains = pkg.synth.plan("cfg2_64ch", nr_channels=4)
    eng = pkg.Engine(fs, decim, 1 << 14, device=0, coalesce_samples=1 << 16)
    for o, g in zip(offs, gains):
        eng.add_channel(int(o), taps, float(g))
    eng.commit()
    room0 = lib.mfm_engine_input_room(eng.h)
    assert room0 >= (1 << 16)
    assert eng.push(np.zeros(2 * 5000, np.int16)) == 0
    r1 = lib.mfm_engine_input_room(eng.h)
    assert room0 - 5000 <= r1 <= room0                    # gathered, or launched at once (the device was idle): then only the carried tail is there
    eng.sync()
    while eng.fetch() is not None:
        pass
    assert lib.mfm_engine_input_room(eng.h) >= room0 - 5000
    eng.close()
