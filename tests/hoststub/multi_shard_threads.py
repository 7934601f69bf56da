"""Run by tests/test_group.py in a fresh process with the fake librccl.so first in LD_LIBRARY_PATH: the advisor's scenario of
round 2 on real engines - a 2-shard group (one device listed twice), one thread pushing small blocks, another one fetching at
the same time.  Round 2's mfm_group_fetch could see shard 0's block before shard 1's submit and fail the receiver with "shards
out of step"; now fetch must only ever report "nothing yet" or a whole block, and the PCM must be the oracle's."""
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package  # noqa: E402
import oracle_lib as ora  # noqa: E402

pkg = load_package()
b = pkg.binding
fs, decim, taps, offs, gains = pkg.synth.plan("cfg2_64ch", nr_channels=10)
grp = b.Group(fs, decim, 4096, devices=(0, 0), flags=b.MFM_F_GROUP_SHARED_DEVICE, exchange=b.MFM_X_RCCL_ALLGATHER)
for o, g in zip(offs, gains):
    grp.add_channel(int(o), taps, float(g))
grp.commit()
nblocks, m = 400, 2048
iq = pkg.synth.random_iq(nblocks * m, seed=77)
parts, errors, done = [], [], threading.Event()


def consumer():
    try:
        while True:
            got = grp.fetch()
            if got is not None:
                parts.append(got)
            elif done.is_set():
                if grp.fetch() is None:
                    return
    except Exception as e:  # MfmError "shards out of step" was the failure of round 2
        errors.append(repr(e))


t = threading.Thread(target=consumer)
t.start()
busy = 0
for k in range(nblocks):
    while grp.push(iq[k * m:(k + 1) * m]) == b.MFM_E_BUSY:
        busy += 1
grp.sync()
done.set()
t.join(timeout=120)
assert not t.is_alive() and not errors, errors
parts.sort(key=lambda p: p[0])
pcm = np.concatenate([p[1] for p in parts], axis=1)
grp.close()
cre = np.stack([ora.make_taps(taps, int(o), fs, float(g))[0] for o, g in zip(offs, gains)])
cim = np.stack([ora.make_taps(taps, int(o), fs, float(g))[1] for o, g in zip(offs, gains)])
incr = np.stack([ora.rot_incr(int(o), fs, decim) for o in offs])
ref, _ = ora.run_channels(iq, cre, cim, incr, decim, threads=8)
assert pcm.shape == ref.shape and np.array_equal(pcm, ref), (pcm.shape, ref.shape)
print(f"threads ok: {len(parts)} blocks fetched concurrently with {nblocks} pushes ({busy} busy retries)")
