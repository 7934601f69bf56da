"""World-size-2 test of the N>1 plumbing on CPU (gloo): channel partition, in-place broadcast of the
wideband block from the ingest rank, per-rank processing of its own channel range, max-over-ranks
timing.  No GPU here, so a rank's "engine" is a stub: per channel and block, a checksum of the block's bytes
keyed by the channel's offset (the HIP engine's arithmetic is the GPU tests' business, and the oracle is not a
stand-in for the product).  What is under test is the plumbing: shard ranges tile the channel set, every rank sees
the ingest rank's bytes in its own buffer through each exchange algorithm, and the concatenated shard outputs equal
what one process computes over all channels."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_ranges_tile_the_channel_set(pkg):
    for n in (1, 2, 63, 64, 65, 1024, 2048):
        for world in (1, 2, 3, 4, 8):
            got = []
            for r in range(world):
                lo, hi = pkg.dist.shard_range(n, r, world)
                got.extend(range(lo, hi))
                assert 0 <= hi - lo <= -(-n // world)
            assert got == list(range(n))
    with pytest.raises(ValueError):
        pkg.dist.shard_range(8, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _stub_engine(offset_hz, iq):
    """what a rank 'demodulates' for one channel from one block: 16 running checksums of the block's samples, keyed by
    the channel (any byte of the block that differs from the ingest rank's changes them)"""
    x = iq.astype(np.int64).reshape(-1)
    w = (np.arange(x.size, dtype=np.int64) * 2654435761 + offset_hz) % 65521
    parts = np.array_split(x * w, 16)
    return np.array([int(p.sum() % 32749) for p in parts], dtype=np.int16)


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from __graft_entry__ import load_package
    pkg = load_package()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        fs, decim, taps, offs, gains = pkg.synth.plan("cfg2_64ch", nr_channels=10)
        lo, hi = pkg.dist.shard_range(len(offs), rank, world)
        block, nblocks = 20000, 3
        chans = [int(o) for o in offs[lo:hi]]
        outs = [[] for _ in chans]
        buf = torch.zeros(2 * block + 64, dtype=torch.int16)
        for b in range(nblocks):
            view = buf[32: 32 + 2 * block]  # a sub-view, like the engine's [tail | block] buffer
            if rank == 0:
                iq = pkg.synth.synth_iq(block, fs, offs[:3], seed=100 + b)
                view.copy_(torch.from_numpy(iq.reshape(-1)))
            else:
                view.fill_(-1)
            if b == 0:
                pkg.dist.broadcast_block(view, src=0)
            elif b == 1:
                pkg.dist.BlockExchange(src=0, algo="scatter_allgather").run(view)
            else:
                ex = pkg.dist.BlockExchange(src=0, algo="auto")
                assert ex.choose(view, sync=lambda: None, iters=1) in pkg.dist.BlockExchange.ALGOS
                assert set(ex.timings) == set(pkg.dist.BlockExchange.ALGOS)
                if rank != 0:
                    view.fill_(-1)  # choose() already delivered the block: make sure run() does so again
                ex.run(view)
            got = view.numpy().reshape(-1, 2)
            for k, off in enumerate(chans):
                outs[k].append(_stub_engine(off, got))
        t = pkg.dist.max_over_ranks(1.0 + rank)
        assert t == float(world)
        pcm = np.stack([np.concatenate(o) for o in outs]) if chans else np.zeros((0, 0), np.int16)
        np.save(os.path.join(out_dir, f"pcm_{rank}.npy"), pcm)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_ranks_exchange_the_block_and_demodulate_their_shards(pkg, ora, tmp_path, world):
    """world 2, and 8 - the node the north star names (10 channels over 8 ranks: shards of two and of one channel)"""
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    parts = [np.load(tmp_path / f"pcm_{r}.npy") for r in range(world)]
    got = np.concatenate(parts, axis=0)
    # single-process reference over all channels
    fs, decim, taps, offs, gains = pkg.synth.plan("cfg2_64ch", nr_channels=10)
    blocks = [pkg.synth.synth_iq(20000, fs, offs[:3], seed=100 + b) for b in range(3)]
    ref = np.stack([np.concatenate([_stub_engine(int(o), blk) for blk in blocks]) for o in offs])
    assert got.shape == ref.shape and np.array_equal(got, ref)
