#!/usr/bin/env python3
"""bench.py -- the headline metric of BASELINE.json on MI355X.

A "step" is one pass of the PEAQ hot path (basic model) over one batch of
synthetic (ref, test) pairs that already sit in HBM: BASELINE.json configs[1],
"Basic PEAQ, 4096 synthetic 48 kHz stereo 10 s (ref,test) pairs batched on 1
MI355X".  With --gpus N every rank (one process per GPU, launched by
torch.distributed.run) processes its own 4096 pairs -- pairs are independent,
there is no data-path collective; the only communication is a
gather of the per-pair ODG scalars over RCCL after the timed region.

Prints ONE JSON line (rank 0).  `value` = stereo 2048-sample frame-pairs per
second over all GPUs; `roofline` prices the dominant kernel (the FFT ear-model
front end) against HBM with the ALGORITHMIC 16 384 B per stereo frame-pair of
SURVEY.md 8(d), from HIP events recorded around that kernel's launches on its
stream; `cpu_baseline` times the reference element itself (oracle/_ref, built
from the reference's sources) -- or the C oracle if that binary is absent -- on
a bounded sample of the same seeded pairs on the host cores.
"""
import argparse
import json
import os
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

ALGO_BYTES_PER_FRAME_PAIR = 16384     # 1024 new samples x 2 ch x 2 signals x 4 B  
HBM_PEAK_GBS = 8000.0                 # MI355X_MICROARCH.md: 8 TB/s
FLOP_PER_FRAME_PAIR = 0.45e6          # SURVEY.md 8(d), reported alongside
FP64_VECTOR_PEAK_TFLOPS = 78.6


def cpu_baseline(n_samples, channels, seed0, advanced=False, budget_s=12.0):
    """frame-pairs/s of the reference C path on ONE host core, bounded sample."""
    cores = 1
    ref_bin = ROOT / "oracle" / "_ref" / "ref_harness"
    env = dict(os.environ)
    # one 10 s stereo pair is ~0.055 s on the reference element, ~0.3 s on the oracle
    pairs = max(2, int(budget_s / (0.75 if advanced else 0.055)))
    try:
        if ref_bin.exists():
            out = subprocess.run([str(ref_bin), "time", str(int(advanced)), str(channels), str(seed0), str(pairs), str(n_samples)],
                                 capture_output=True, text=True, timeout=240, env=env)
            if out.returncode == 0:
                d = json.loads(out.stdout.strip().splitlines()[-1])
                return dict(value=d["frame_pairs_per_s"], unit="frame-pairs/s", cores=cores, kind="reference",
                            sample=f"{d['pairs']} of the same seeded 10 s stereo pairs ({d['frame_pairs']} frame-pairs) "
                                   f"through the reference `peaq` element (oracle/_ref), 1 thread, "
                                   f"{d['seconds']:.1f} s; host has {os.cpu_count()} cores")
    except Exception:
        pass
    pairs = max(2, int(budget_s / (1.5 if advanced else 0.3)))
    cli = ROOT / "oracle" / "oracle_cli"
    if not cli.exists():
        subprocess.run(["make", "-C", str(ROOT / "oracle"), "oracle_cli"], capture_output=True)
    out = subprocess.run([str(cli), "time", str(int(advanced)), str(channels), str(seed0), str(pairs), str(n_samples)],
                         capture_output=True, text=True, timeout=180)
    d = json.loads(out.stdout.strip().splitlines()[-1])
    return dict(value=d["frame_pairs_per_s"], unit="frame-pairs/s", cores=cores, kind="port",
                sample=f"{d['pairs']} of the same seeded 10 s stereo pairs ({d['frame_pairs']} frame-pairs) through "
                       f"oracle/peaq_oracle.c, 1 thread, {d['seconds']:.1f} s; host has {os.cpu_count()} cores")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--pairs", type=int, default=4096, help="pairs per GPU (configs[1]: 4096)")
    ap.add_argument("--seconds", type=float, default=10.0, help="length of each pair")
    ap.add_argument("--channels", type=int, default=2)
    ap.add_argument("--advanced", action="store_true", help="configs[2] instead of configs[1]")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import gstpeaq_amd

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)      # "nccl" is RCCL on ROCm

    ctx = gstpeaq_amd.Context(local_rank)
    n_samples = int(round(args.seconds * 48000))
    seed0 = 1 + rank * args.pairs                           # rank r owns pairs [r*P, (r+1)*P)
    ref, test = gstpeaq_amd.synth_fill(ctx, seed0, args.pairs, args.channels, n_samples, device=dev)
    results = torch.empty((args.pairs, 16), dtype=torch.float64, device=dev)
    torch.cuda.synchronize(dev)

    def step():
        gstpeaq_amd.batch_run(ctx, args.advanced, ref, test, results=results, sync=False)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    if dist:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize(dev)
    if dist:
        dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    timing = ctx.last_timing()                              # HIP events of the last step, on the launch stream

    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if dist:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())

    # the one collective of the path: gather the per-pair result records
    from gstpeaq_amd import parallel
    frame_pairs_rank = float(results[:, 14].sum().item())
    gathered = parallel.gather_results(results, world, dist)
    odg = gathered[:, 12]
    frame_pairs_all = float(gathered[:, 14].sum().item())

    if rank == 0:
        value = frame_pairs_all * args.steps / elapsed
        fe_s = timing["frontend_ms"] * 1e-3
        achieved = frame_pairs_rank * ALGO_BYTES_PER_FRAME_PAIR / fe_s / 1e9 if fe_s > 0 else 0.0
        # the bound that actually applies: FP64 vector issue.  VALU instructions per wave
        # of the dominant kernel come from a rocprofv3 --pmc pass (profiles/r01_basic_1024_pmc_mix.json);
        # one wave64 instruction occupies its SIMD for 4 cycles, an MI355X has 256 CUs x 4 SIMDs at 2.4 GHz
        valu_frac = None
        mix = ROOT / "profiles" / "r01_basic_1024_pmc_mix.json"
        if mix.exists() and not args.advanced:
            try:
                k = next(v for n, v in json.loads(mix.read_text()).items() if "frontend_kernel<109>" in n)
                per_wave = k["SQ_INSTS_VALU"]["avg"] / k["SQ_WAVES"]["avg"]
                waves = frame_pairs_rank * args.channels * 2          # one wave per (frame, channel, signal)
                valu_frac = waves * per_wave * 4 / (1024 * 2.4e9) / fe_s
            except Exception:
                valu_frac = None
        traffic = None
        prof = ROOT / "profiles" / "pmc_frontend.json"     # written from a rocprofv3 --pmc pass (see profiles/README.md)
        if prof.exists():
            try:
                traffic = json.loads(prof.read_text()).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "2048-sample stereo ref/test frame-pairs/sec (basic PEAQ, ear model -> MOVs -> ODG)"
                      if not args.advanced else "FFT frame-pairs/sec (advanced PEAQ incl. filter-bank blocks)",
            "value": value,
            "unit": "frame-pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"{'Advanced' if args.advanced else 'Basic'} PEAQ, {args.pairs} synthetic 48 kHz "
                                   f"{'stereo' if args.channels == 2 else 'mono'} {args.seconds:g} s (ref,test) pairs "
                                   f"per GPU, inputs resident in HBM (BASELINE.json configs[{2 if args.advanced else 1}])",
                       "pairs_per_gpu": args.pairs, "frame_pairs_per_pair": frame_pairs_rank / args.pairs,
                       "parallelism": f"pairs sharded over {world} GPU(s), no data-path collective"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "frontend_kernel<55>" if args.advanced else "frontend_kernel<109>", "launches": timing["frontend_launches"],
                         "avg_launch_ms": timing["frontend_ms"] / max(timing["frontend_launches"], 1),
                         "algorithmic_bytes_per_launch": frame_pairs_rank * ALGO_BYTES_PER_FRAME_PAIR
                                                         / max(timing["frontend_launches"], 1),
                         "compute_frac_fp64_vector": value / world * FLOP_PER_FRAME_PAIR / (FP64_VECTOR_PEAK_TFLOPS * 1e12),
                         "valu_issue_frac": valu_frac,
                         "backend_ms": timing["backend_ms"], "fb_ms": timing["fb_ms"],
                         "step_ms_events": timing["total_ms"]},
            "odg_mean": float(odg[~torch.isnan(odg)].mean().item()),
            "odg_nan": int(torch.isnan(odg).sum().item()),
        }
        if args.advanced and timing["fb_launches"]:
            # configs[2]: the dominant kernel is the filter bank (fb_bank_kernel), a folded FIR bank on the
            # matrix cores.  Algorithmic work as the reference counts it (fbearmodel.c:404-434): per tap
            # pair two additions and two multiply-adds, 10 914 tap pairs per sub-sample, 6 sub-samples per
            # 192-sample block; against the FP64 matrix peak (= FP64 vector peak, 78.6 TFLOP/s).
            blocks = float(gathered[: args.pairs, 15].sum().item()) if world > 1 else float(results[:, 15].sum().item())
            flops = blocks * args.channels * 2 * 6 * 10914 * 6
            fb_s = timing["fb_ms"] * 1e-3
            tf = flops / fb_s / 1e12
            line["roofline"] = {"bound": "mfma", "achieved": tf, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                                "frac": tf / FP64_VECTOR_PEAK_TFLOPS, "traffic": None, "kernel": "fb_bank_kernel",
                                "launches": timing["fb_launches"],
                                "avg_launch_ms": timing["fb_ms"] / timing["fb_launches"],
                                "algorithmic_flop_per_launch": flops / timing["fb_launches"],
                                "frontend_ms": timing["frontend_ms"], "backend_ms": timing["backend_ms"],
                                "fb_ms": timing["fb_ms"], "step_ms_events": timing["total_ms"]}
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(n_samples, args.channels, seed0, args.advanced)
        print(json.dumps(line))
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
