"""Multi-GPU plumbing for the channel-sharded multifm engine (one process per GPU).

Channels are independent given the same wideband input (the reference runs them as independent
threads, multifm/receiver.c:89-95), so the only exchange step is handing every rank the wideband
block: a broadcast from the ingest rank (RCCL over xGMI on the GPUs, gloo in the CPU tests).
Outputs are disjoint per channel - no gather, no reduction.
"""


def shard_range(nr_channels, rank, world):
    """Contiguous range [lo, hi) of channels owned by `rank`; sizes differ by at most one."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError("bad rank/world")
    base, extra = divmod(nr_channels, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def broadcast_block(view, src=0, group=None, async_op=False):
    """Deliver the ingest rank's wideband block (an int16 tensor view of the engine's input buffer) to
    every rank, in place."""
    import torch
    import torch.distributed as dist
    # neither RCCL nor gloo has a 16-bit integer type; a broadcast only moves bytes
    return dist.broadcast(view.view(torch.uint8), src=src, group=group, async_op=async_op)


def max_over_ranks(seconds, device="cpu"):
    """Step time of the job = slowest rank."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
