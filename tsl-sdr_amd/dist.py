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


class BlockExchange:
    """The exchange step of the channel-sharded engine: the ingest rank's wideband block reaches every rank, in place.

    Two ways to do it, same bytes in the same places afterwards:
      * "broadcast"          one RCCL broadcast (rings of single xGMI links);
      * "scatter_allgather"  the ingest rank sends rank r the r-th 1/N of the block (N - 1 different links at once),
                             then an all-gather fills in the rest (every link of the node busy): the classic
                             large-message broadcast, worth it where links are point to point.
    "auto" times both on the real buffer once (choose()) and keeps the faster; the decision is taken on max-reduced
    times, so every rank takes the same one.  bench.py's default at N > 1 is "scatter_allgather" (round 4): a broadcast
    delivers the block at one link's rate per GPU whatever N, the scatter + all-gather uses every link (DESIGN.md section
    7) - and a fixed choice keeps every rank on the same sequence of collectives whatever happens.  A collective that fails
    is NOT caught anywhere here: after a failed RCCL operation the communicator is not usable for a fallback, and ranks that
    disagree about which collective comes next hang; the error ends the process (and with it the job)."""

    ALGOS = ("broadcast", "scatter_allgather")

    def __init__(self, src=0, algo="auto", group=None):
        import torch.distributed as dist
        self.src, self.group = src, group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.algo = algo if self.world > 1 else "broadcast"
        self.timings = None

    def _scatter_allgather(self, u8):
        import torch.distributed as dist
        n = u8.numel()
        if n % self.world:
            return dist.broadcast(u8, src=self.src, group=self.group)
        chunks = list(u8.chunk(self.world))
        if self.rank == self.src:
            ops = [dist.P2POp(dist.isend, chunks[r], r, self.group) for r in range(self.world) if r != self.src]
        else:
            ops = [dist.P2POp(dist.irecv, chunks[self.rank], self.src, self.group)]
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        if dist.get_backend(self.group) == "gloo":
            # gloo gathers into a list of tensors; the chunks are views of the block, so this is in place as well
            got = [c.clone() for c in chunks]
            dist.all_gather(got, chunks[self.rank].clone(), group=self.group)
            for c, g in zip(chunks, got):
                c.copy_(g)
        else:
            dist.all_gather_into_tensor(u8, chunks[self.rank], group=self.group)
        return None

    def run(self, view, algo=None):
        import torch
        import torch.distributed as dist
        u8 = view.view(torch.uint8)  # neither RCCL nor gloo has a 16-bit integer type; the exchange only moves bytes
        algo = algo or self.algo
        if algo == "auto":
            algo = "broadcast"  # until choose() has run
        if algo == "scatter_allgather":
            return self._scatter_allgather(u8)
        return dist.broadcast(u8, src=self.src, group=self.group)

    def choose(self, view, sync, iters=3):
        """Time both algorithms on `view` (sync() must drain the device) and keep the faster one."""
        import time
        import torch.distributed as dist
        if self.world == 1 or self.algo != "auto":
            return self.algo
        dev = view.device
        res = {}
        for algo in self.ALGOS:
            # both forms use operations every backend in use here has (broadcast, batched send / receive, all-gather: RCCL
            # and gloo); an error is a real one and propagates - every rank runs the same collectives in the same order
            self.run(view, algo)
            sync()
            dist.barrier(group=self.group)
            t0 = time.perf_counter()
            for _ in range(iters):
                self.run(view, algo)
            sync()
            res[algo] = max_over_ranks((time.perf_counter() - t0) / iters, device=dev)
        self.timings = res
        self.algo = min(res, key=res.get)
        return self.algo


def max_over_ranks(seconds, device="cpu"):
    """Step time of the job = slowest rank."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
