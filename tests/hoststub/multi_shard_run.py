"""Run by tests/test_group.py in a fresh process whose LD_LIBRARY_PATH starts with a directory holding the fake librccl.so
(tests/hoststub/fake_rccl.cpp): a device group of several shards on ONE GPU (MFM_F_GROUP_SHARED_DEVICE), blocks in int16
and 8-bit formats through the chosen exchange mode, PCM of all shards against the oracle.  argv: nr_shards mode nr_channels [coalesce_samples [gather]]."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package  # noqa: E402
import oracle_lib as ora  # noqa: E402

pkg = load_package()
b = pkg.binding
S, mode, nch = int(sys.argv[1]), sys.argv[2], int(sys.argv[3])
coalesce = int(sys.argv[4]) if len(sys.argv) > 4 else 0
gather = len(sys.argv) > 5 and sys.argv[5] == "gather"
fs, decim, taps, offs, gains = pkg.synth.plan("cfg2_64ch", nr_channels=nch)
grp = b.Group(fs, decim, 1 << 16, devices=(0,) * S, flags=b.MFM_F_GROUP_SHARED_DEVICE | (b.MFM_F_GATHER if gather else 0),
              coalesce_samples=coalesce,
              exchange={"rccl": b.MFM_X_RCCL, "allgather": b.MFM_X_RCCL_ALLGATHER, "auto": b.MFM_X_AUTO}[mode])
for o, g in zip(offs, gains):
    grp.add_channel(int(o), taps, float(g))
grp.commit()
assert grp.nr_shards == min(S, nch), grp.nr_shards
covered = []
for s in range(grp.nr_shards):
    lo, n, dev = grp.shard_info(s)
    covered.extend(range(lo, lo + n))
assert covered == list(range(nch))
rng = np.random.RandomState(S * 100 + nch)
blocks = [(rng.randint(-32768, 32768, size=(m, 2)).astype(np.int16), 0) for m in (65536, 30001, 7, 96, 50000)]
blocks += [(rng.randint(0, 256, size=(m, 2)).astype(np.uint8), 3) for m in (40000, 4097, 65536)]
blocks += [(rng.randint(-32768, 32768, size=(33333, 2)).astype(np.int16), 0)]
iq, parts = [], []
for blk, fmt in blocks:
    iq.append(blk if fmt == 0 else ora.unpack_bytes(blk, fmt).reshape(-1, 2))
    while grp.push(blk, fmt) == b.MFM_E_BUSY:
        parts.append(grp.fetch()[1])
while True:
    rc = grp.flush()  # a coalescing group may hold accepted blocks it has not launched
    while True:
        got = grp.fetch()
        if got is None:
            break
        parts.append(got[1])
    if rc == 0:
        break
grp.sync()
uses, nblk, moved = grp.exchange_info()
st = [grp.stats(s) for s in range(grp.nr_shards)]
grp.close()
iq = np.concatenate(iq)
cre = np.stack([ora.make_taps(taps, int(o), fs, float(g))[0] for o, g in zip(offs, gains)])
cim = np.stack([ora.make_taps(taps, int(o), fs, float(g))[1] for o, g in zip(offs, gains)])
incr = np.stack([ora.rot_incr(int(o), fs, decim) for o in offs])
ref, _ = ora.run_channels(iq, cre, cim, incr, decim, threads=8)
pcm = np.concatenate(parts, axis=1)
assert pcm.shape == ref.shape, (pcm.shape, ref.shape)
bad = np.argwhere(pcm != ref)
assert len(bad) == 0, f"{len(bad)} PCM samples differ, first at (chan, n) = {bad[0]}"
assert uses and nblk == len(blocks) and moved > 0, (uses, nblk, moved)
if coalesce:
    # the shards launch or defer together: same launches, same positions
    assert len(set(s["launches"] for s in st)) == 1 and len(set(s["submits"] for s in st)) == 1, st
    if gather:
        assert st[0]["launches"] < st[0]["submits"], (st[0]["launches"], st[0]["submits"])
print(f"multi-shard ok: {grp.nr_shards} shards, mode {mode}, {nch} channels, {pcm.shape[1]} outputs, {moved} bytes exchanged, "
      f"8-bit launches per shard {[s['launches_8bit'] for s in st]}")
