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


def waves(n_pairs, wave_pairs):
    """[(first pair, count)] of the waves a share of n_pairs is consumed in"""
    return [(w0, min(wave_pairs, n_pairs - w0)) for w0 in range(0, n_pairs, wave_pairs)]


def run_waves(ctx, advanced, seed0, n_pairs, wave_pairs, ref, test, results, playback_level=92.0):
    """One pass over a GPU's share [seed0, seed0 + n_pairs) that does not fit in HBM at once
    (BASELINE.json configs[3]: 32 768 pairs = 252 GB per GPU): wave w is generated on the device
    into the resident buffers `ref`/`test` ([>= wave_pairs, n_samples, channels]), run through the
    batch path, and only its 128-byte result records are kept in results[w0 : w0 + count].
    Returns the seconds spent inside the batch calls (device-synchronised on both sides of each);
    generation is outside of it ("generation time excluded from the metric")."""
    import time
    import torch
    from . import capi
    dev = ref.device
    n_samples, channels = ref.shape[1], ref.shape[2]
    timed = 0.0
    for w0, cnt in waves(n_pairs, wave_pairs):
        capi.synth_fill(ctx, seed0 + w0, cnt, channels, n_samples, out=(ref, test))
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        capi.batch_run(ctx, advanced, ref[:cnt], test[:cnt], results=results[w0:w0 + cnt],
                       playback_level=playback_level, sync=False)
        torch.cuda.synchronize(dev)
        timed += time.perf_counter() - t0
    return timed


def gather_results(local, world, dist=None):
    """local: tensor [n_local, k] of result records -> [n_total, k] on every rank,
    in pair order.  Shards may differ in size by one."""
    import torch
    if dist is None:                      # single process, no communicator
        return local
    if dist.get_backend() == "gloo":      # CPU communicator (tests; several ranks on ONE GPU): through host memory
        local = local.cpu()
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
