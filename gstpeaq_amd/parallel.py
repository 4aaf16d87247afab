"""Multi-GPU layout of the batch path: pairs are closed
computations, so rank r of W simply owns a contiguous block of pair indices and
nothing is exchanged while the kernels run.  The one collective is the gather
of the per-pair result records (104-128 B each) at the end -- RCCL over xGMI on
the GPU box ("nccl" backend), gloo in the CPU tests."""


def shard(n_total, rank, world):
    """contiguous block [lo, hi) of pair indices owned by `rank`"""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_results(local, world, dist=None):
    """local: tensor [n_local, k] of result records -> [n_total, k] on every rank,
    in pair order.  Shards may differ in size by one."""
    import torch
    if world == 1 or dist is None:
        return local
    n = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    width = max(sizes)
    pad = torch.zeros((width, local.shape[1]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    return torch.cat([p[:s] for p, s in zip(parts, sizes)])
