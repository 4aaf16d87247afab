#!/usr/bin/env python3
"""PCIe-inclusive batch rate: the (ref, test) pairs start in PINNED HOST memory and are copied to
HBM chunk by chunk on a copy stream while the previous chunk is being processed (double
buffering).  This is the rate a host-fed batch caller sees; it is NOT bench.py's `value` (inputs
resident in HBM).  Prints one JSON line."""
import argparse
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=4096)
    ap.add_argument("--chunk", type=int, default=512)
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--advanced", action="store_true")
    args = ap.parse_args()
    import torch
    import gstpeaq_amd
    dev = torch.device("cuda", 0)
    ctx = gstpeaq_amd.Context(0)
    n = int(args.seconds * 48000)
    # one chunk of seeded pairs, generated on the device and parked in pinned host memory; the
    # same host chunk is fed repeatedly (the copies are real, the content does not matter)
    ref_d, test_d = gstpeaq_amd.synth_fill(ctx, 1, args.chunk, 2, n, device=dev)
    ref_h = torch.empty(ref_d.shape, dtype=ref_d.dtype, pin_memory=True).copy_(ref_d)
    test_h = torch.empty(test_d.shape, dtype=test_d.dtype, pin_memory=True).copy_(test_d)
    bufs = [(ref_d, test_d), (torch.empty_like(ref_d), torch.empty_like(test_d))]
    results = [torch.empty((args.chunk, 16), dtype=torch.float64, device=dev) for _ in range(2)]
    copy_s, comp_s = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    copied = [torch.cuda.Event() for _ in range(2)]
    done = [torch.cuda.Event() for _ in range(2)]
    n_chunks = args.pairs // args.chunk

    def run():
        for i in range(n_chunks):
            b = i & 1
            with torch.cuda.stream(copy_s):
                if i >= 2:
                    copy_s.wait_event(done[b])                   # the buffer is free again
                bufs[b][0].copy_(ref_h, non_blocking=True)
                bufs[b][1].copy_(test_h, non_blocking=True)
                copied[b].record(copy_s)
            with torch.cuda.stream(comp_s):
                comp_s.wait_event(copied[b])
                gstpeaq_amd.batch_run(ctx, args.advanced, bufs[b][0], bufs[b][1], results=results[b], sync=False,
                                      stream=comp_s)
                done[b].record(comp_s)
        torch.cuda.synchronize(dev)

    run()                                                        # warm-up
    t0 = time.perf_counter()
    run()
    dt = time.perf_counter() - t0
    frames = float(results[0][:, 14].sum().item()) * n_chunks
    gb = 2 * ref_h.numel() * 4 * n_chunks / 1e9
    print(json.dumps(dict(pairs=args.pairs, chunk=args.chunk, wall_s=dt, frame_pairs_per_s=frames / dt,
                          h2d_GBps=gb / dt, mode="advanced" if args.advanced else "basic",
                          note="pinned host memory -> HBM over PCIe, double buffered, overlapped with the kernels")))


if __name__ == "__main__":
    main()
