#!/usr/bin/env python3
"""tools/parity_soak.py -- the whole of BASELINE.json configs[1] and configs[2] against the REAL reference element.

Every pair of the bench workload (4096 seeded 48 kHz stereo 10 s pairs by default) goes through the HIP path once per
version (basic; advanced on the engine's default FP64 arithmetic) and through `oracle/_ref/ref_harness` -- the
reference's own sources compiled here, one process per usable host core, each on its own block of seeds.  Printed: one
JSON object with, per version, max and 99th percentile of |delta ODG| and |delta DI|, the largest relative difference
of every MOV, NaN mismatches, and for the discretely gated MOVs (a threshold decides whether a frame counts:
Bandwidth, RelDistFrames, ADB, MFPD) the number of pairs on which they differ beyond rounding.

  python tools/parity_soak.py [--pairs 4096] [--advanced-pairs 4096] [--seconds 10] > profiles/rNN_parity_soak.json

Test infrastructure: the reference is only ever the thing compared WITH (never timed here, never on the product path).
About 0.36 s (basic) / 0.75 s (advanced) of one core per pair, generation included."""
import argparse
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402  (cpu_records, result_deltas)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=4096)
    ap.add_argument("--advanced-pairs", type=int, default=None, help="default: --pairs")
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--channels", type=int, default=2)
    ap.add_argument("--seed", type=int, default=1, help="pair i carries seed SEED + i (bench.py: 1)")
    ap.add_argument("--batch", type=int, default=4096, help="pairs resident on the GPU at a time")
    args = ap.parse_args()
    import numpy as np
    import torch
    import gstpeaq_amd

    n_samples = int(round(args.seconds * 48000))
    ctx = gstpeaq_amd.Context(0)
    assert ctx.fir_mode() == "f64"
    tool, kind, what = bench._cpu_tool()
    out = {"workload": f"{args.pairs} synthetic 48 kHz {'stereo' if args.channels == 2 else 'mono'} {args.seconds:g} s pairs, "
                       f"seeds {args.seed} .. {args.seed + args.pairs - 1} (bench.py's)",
           "reference": what, "reference_kind": kind, "library": str(gstpeaq_amd.library_path())}
    for advanced, n in ((0, args.pairs), (1, args.advanced_pairs if args.advanced_pairs is not None else args.pairs)):
        if n <= 0:
            continue
        rows = np.empty((n, 16))
        t0 = time.time()
        for b0 in range(0, n, args.batch):
            nb = min(args.batch, n - b0)
            ref, test = gstpeaq_amd.synth_fill(ctx, args.seed + b0, nb, args.channels, n_samples)
            res = torch.empty((nb, 16), dtype=torch.float64, device=ref.device)
            gstpeaq_amd.batch_run(ctx, advanced, ref, test, results=res, sync=True)
            rows[b0:b0 + nb] = res.cpu().numpy()
            del ref, test, res
        gpu_s = time.time() - t0
        cpu = bench.cpu_records(n_samples, args.channels, args.seed, bool(advanced), n, timeout=6000)
        d = bench.result_deltas(rows, cpu, bool(advanced))
        d.update(pairs=n, gpu_seconds_incl_generation=round(gpu_s, 2), reference_seconds=round(cpu["seconds"], 1),
                 reference_processes=cpu["cores"], odg_mean_gpu=float(np.nanmean(rows[:, 12])),
                 odg_min_gpu=float(np.nanmin(rows[:, 12])), odg_max_gpu=float(np.nanmax(rows[:, 12])),
                 frame_pairs=float(rows[:, 14].sum()))
        out["advanced" if advanced else "basic"] = d
        print(("advanced" if advanced else "basic"), json.dumps(d), file=sys.stderr, flush=True)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
