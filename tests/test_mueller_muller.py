"""BASELINE.json configs[3] names a "mueller_muller slicer": pager/mueller_muller.c (mm_init / mm_process).  The live
decoder does not call it and the reference's own test (pager/test/test_mueller_muller.c) needs a capture file that
is not in the tree, so the oracle restatement is PARITY UNPINNED; the tests below are that test's procedure on a
synthetic 1200-baud POCSAG transmission at 25 kHz: same gains (:85-88), same slicing of one long buffer (:108-113),
same consumer (decision > 0 ? 0 : 1 into a shift register, sync word when fewer than 4 bits differ, :118-121)."""
import numpy as np
import pytest

KW, KM = 0.0001, 0.000004
SPB = np.float32(25000.0) / np.float32(1200.0)
MARGIN = np.float32(0.05)
SYNC = 0x7CD215D8


def _count_syncs(decisions):
    """Sync words 544 decisions apart, counted from the first one (test_mueller_muller.c:118-127); what matches by
    accident in the noise behind the transmission is not part of the chain."""
    shr, pos = 0, []
    for i, d in enumerate(decisions):
        shr = ((shr << 1) | (0 if d > 0 else 1)) & 0xFFFFFFFF
        if bin(SYNC ^ shr).count("1") < 4:
            pos.append(i)
    n = 0
    for p in pos:
        if p == pos[0] + 544 * n:
            n += 1
    return n


def _transmission(pkg, seed, lead, nbatches_msgs=3):
    sy = pkg.synth
    msgs = [(0x12345 + seed, 3, 2, sy.pocsag_alpha_words("MUELLER AND MULLER %d" % seed)),
            (0x00777, 5, 0, sy.pocsag_numeric_words("0123-456 9")),
            (0x3FFF0 + seed, 0, 3, sy.pocsag_alpha_words("The quick brown fox jumps over the lazy dog " * 2))][:nbatches_msgs]
    batches = sy.pocsag_batches(msgs)
    pcm = sy.pocsag_pcm(sy.pocsag_bits(batches), 1200, noise=900, lead=lead, trail=4000, seed=seed, rate=25000)
    return pcm, len(batches)


def _oracle_run(ora, buf, n, per_iter):
    mm = ora.MuellerMuller(KW, KM, float(SPB), float(SPB - MARGIN), float(SPB + MARGIN))
    out, off = [], 0
    while off < n:
        it = min(per_iter, n - off)
        out.append(mm.process(buf, off, it))
        off += it
    return out


def test_oracle_mueller_muller_finds_every_sync_word(pkg, ora):
    pcm, nb = _transmission(pkg, 1, lead=3000)
    buf = np.concatenate([pcm, np.zeros(8, np.int16)])  # the loop may look one sample past a slice (:66-67)
    per_iter = int(256 * float(SPB))                      # test_mueller_muller.c:100
    dec = np.concatenate(_oracle_run(ora, buf, pcm.size, per_iter))
    assert _count_syncs(dec) == nb                        # :135, on their capture: 9
    assert abs(dec.size - pcm.size / float(SPB)) < 8      # one decision per bit period
    # decisions are samples of the input, taken in order
    assert np.all(np.isin(dec[:50], pcm))


@pytest.mark.gpu
def test_gpu_mueller_muller_matches_oracle(pkg, ora):
    """Seven channels with different timing, sliced like the reference's test; decisions and counts identical."""
    nch = 7
    chans, nbs = [], []
    for c in range(nch):
        pcm, nb = _transmission(pkg, c, lead=1000 + 977 * c, nbatches_msgs=1 + c % 3)
        chans.append(pcm)
        nbs.append(nb)
    n = max(p.size for p in chans)
    buf = np.zeros((nch, n + 8), np.int16)
    for c, p in enumerate(chans):
        buf[c, :p.size] = p
    per_iter = int(256 * float(SPB))
    gpu = pkg.MuellerMuller(nch, KW, KM, float(SPB), float(SPB - MARGIN), float(SPB + MARGIN), per_iter, device=0)
    got = [[] for _ in range(nch)]
    off = 0
    while off < n:
        it = min(per_iter, n - off)
        # rows are n + 8 apart, so the sample behind the slice is there for the kernel as it is for the oracle
        view = np.ascontiguousarray(buf[:, off:off + it + 1])
        for c, d in enumerate(gpu.process_host(view, it)):
            got[c].append(d)
        off += it
    gpu.close()
    for c in range(nch):
        want = _oracle_run(ora, buf[c], n, per_iter)
        assert len(want) == len(got[c])
        for a, b in zip(want, got[c]):
            assert np.array_equal(a, b)
        assert _count_syncs(np.concatenate(got[c])) == nbs[c]


@pytest.mark.gpu
def test_gpu_mueller_muller_long_block_and_missing_lookahead(pkg, ora):
    """One call over a whole transmission (many LDS windows); without the look-ahead sample the last sample stands
    in for it, on both sides."""
    pcm, nb = _transmission(pkg, 5, lead=12345)
    n = pcm.size
    gpu = pkg.MuellerMuller(1, KW, KM, float(SPB), float(SPB - MARGIN), float(SPB + MARGIN), n, device=0)
    (d,) = gpu.process_host(pcm.reshape(1, -1), n)   # stride == nr_in: no look-ahead
    gpu.close()
    buf = np.concatenate([pcm, pcm[-1:]])            # what "the last sample stands in" means for the oracle
    mm = ora.MuellerMuller(KW, KM, float(SPB), float(SPB - MARGIN), float(SPB + MARGIN))
    assert np.array_equal(mm.process(buf, 0, n), d)
    assert _count_syncs(d) == nb


@pytest.mark.gpu
def test_gpu_mueller_muller_more_channels_than_a_wave(pkg, ora):
    """70 channels (two workgroups, the second one partly empty), each a shifted copy of three transmissions."""
    base = [_transmission(pkg, s, lead=2000 + 500 * s, nbatches_msgs=1 + s % 3)[0] for s in range(3)]
    nch, n = 70, 60000
    buf = np.zeros((nch, n + 1), np.int16)
    rng = np.random.RandomState(3)
    for c in range(nch):
        p = np.roll(np.resize(base[c % 3], n), 137 * c) + rng.randint(-300, 300, n)
        buf[c, :n] = np.clip(p, -32768, 32767)
    buf[:, n] = buf[:, n - 1]
    gpu = pkg.MuellerMuller(nch, KW, KM, float(SPB), float(SPB - MARGIN), float(SPB + MARGIN), n, device=0)
    got = gpu.process_host(buf, n)
    gpu.close()
    for c in range(nch):
        mm = ora.MuellerMuller(KW, KM, float(SPB), float(SPB - MARGIN), float(SPB + MARGIN))
        assert np.array_equal(mm.process(buf[c], 0, n), got[c]), f"channel {c}"


@pytest.mark.gpu
def test_gpu_mueller_muller_ragged_calls(pkg, ora):
    """Calls of 1 .. 5000 samples, with and without the look-ahead sample, on 19 channels (two workgroups, the second
    one partly filled): the chunked ring (1024-sample chunks, rows shorter than one 16-byte vector, the partial
    vector at a row's end) and the carried state against the oracle, call by call."""
    rng = np.random.RandomState(11)
    nch = 19
    base = [_transmission(pkg, 20 + k, lead=300 + 91 * k, nbatches_msgs=1)[0] for k in range(3)]
    total = 60000
    buf = np.zeros((nch, total + 1), np.int16)
    for c in range(nch):
        buf[c, :total] = np.clip(np.roll(np.resize(base[c % 3], total), 53 * c) + rng.randint(-200, 200, total),
                                 -32768, 32767)
    buf[:, total] = buf[:, total - 1]
    sizes = [1, 2, 7, 8, 9, 3, 1023, 1024, 1025, 17, 2048, 4999, 5, 5000, 1, 1, 31, 4096, 2047]
    while sum(sizes) < total - 5000:
        sizes.append(int(rng.randint(1, 5001)))
    gpu = pkg.MuellerMuller(nch, KW, KM, float(SPB), float(SPB - MARGIN), float(SPB + MARGIN), 5000, device=0)
    mms = [ora.MuellerMuller(KW, KM, float(SPB), float(SPB - MARGIN), float(SPB + MARGIN)) for _ in range(nch)]
    off = 0
    for k, it in enumerate(sizes):
        ahead = k % 3 != 0  # every third call hands over exactly nr_in columns: the last sample stands in
        view = np.ascontiguousarray(buf[:, off:off + it + (1 if ahead else 0)])
        got = gpu.process_host(view, it)
        for c in range(nch):
            src = buf[c] if ahead else np.concatenate([buf[c, :off + it], buf[c, off + it - 1:off + it]])
            assert np.array_equal(mms[c].process(src, off, it), got[c]), f"call {k} ({it} samples), channel {c}"
        off += it
    gpu.close()


def test_mueller_muller_create_refuses_what_the_float_position_cannot_hold(pkg):
    """The reference keeps the sample position in a float (mueller_muller.c:58-66, :96); the stage refuses the
    configurations where that float would stop holding whole numbers exactly, or where a step can fall below one
    sample - before it looks for a device, so this runs without one."""
    spb = float(SPB)
    for kw, km, emin, emax, max_in in ((KW, KM, spb - 0.05, spb + 0.05, 1 << 22),        # block too long
                                       (KW, 0.001, spb - 0.05, spb + 0.05, 5000),         # step can reach zero
                                       (float("inf"), KM, spb - 0.05, spb + 0.05, 5000),  # gain not finite
                                       (KW, KM, spb - 0.05, 2.0e6, 5000),                 # step too long
                                       (KW, KM, spb + 0.05, spb - 0.05, 5000)):           # bounds the wrong way round
        with pytest.raises(pkg.MfmError) as e:
            pkg.MuellerMuller(1, kw, km, spb, emin, emax, max_in, device=0)
        assert e.value.code == pkg.binding.MFM_E_INVAL
