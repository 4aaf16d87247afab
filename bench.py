#!/usr/bin/env python3
"""bench.py -- the headline metric of BASELINE.json on MI355X.

A "step" is one pass of the PEAQ hot path (basic model) over one batch of
synthetic (ref, test) pairs.

  --gpus 1 (default)  BASELINE.json configs[1]: 4096 synthetic 48 kHz stereo 10 s
                      pairs, resident in HBM when the timed region starts.
  --gpus N > 1        BASELINE.json configs[3]'s sharding: every GPU owns 32 768
                      pairs (N = 8: 262 144 pairs), a contiguous block of pair
                      indices (gstpeaq_amd.parallel.shard), consumed in WAVES of
                      4096 pairs: 252 GB of input per GPU do not fit beside the
                      workspace, so wave w is generated on the device into the
                      same two buffers (not timed), then run
                      (timed), and only its 128-byte result records are kept.
                      One process per GPU (torch.distributed.run), no data-path
                      collective; after the timed region the
                      records are gathered with one RCCL all_gather.

Prints ONE JSON line (rank 0): `value` = stereo 2048-sample frame-pairs per
second over all GPUs; `roofline` prices the dominant kernel (the FFT ear-model
front end) against HBM with the ALGORITHMIC 16 384 B per stereo frame-pair of
SURVEY.md 8(d), from HIP events recorded around that kernel's launches on its
stream; `cpu_baseline` times the reference element itself (oracle/_ref, built
from the reference's sources; the C oracle if that binary is absent) on one core
and on all host cores; `odg_max_abs_delta` etc. compare the GPU results with the
reference's on the pairs the single-core leg ran (the second half of
BASELINE.json's metric); `advanced` carries configs[2] (advanced model, same
pairs) on the engine's default, the reference's precision -- every stage FP64 --
with the roofline of that engine's filter-bank kernel, and the opt-in
reduced-precision FIR as the labelled sub-object `reduced_precision_f16x3`;
`scaling_reference` is this GPU in the regime every rank of an N > 1 run is in.
"""
import argparse
import json
import math
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
FP32_MATRIX_PEAK_TFLOPS = 157.3        # v_mfma_f32_16x16x4_f32 (MI355X_MICROARCH.md)
FP16_MATRIX_PEAK_TFLOPS = 2516.0       # v_mfma_f32_16x16x32_f16, dense (MI355X_MICROARCH.md: 2.5 PFLOP/s)
CONFIG4_PAIRS_PER_GPU = 32768         # BASELINE.json configs[3]: 262 144 pairs over 8 GPUs
WAVE_PAIRS = 4096


# ----------------------------------------------------------------------------------------------
# CPU baseline (test infrastructure under oracle/, only ever the thing compared WITH)
# ----------------------------------------------------------------------------------------------
def _cpu_tool():
    ref_bin = ROOT / "oracle" / "_ref" / "ref_harness"
    if ref_bin.exists():
        return str(ref_bin), "reference", "the reference `peaq` element (oracle/_ref)"
    cli = ROOT / "oracle" / "oracle_cli"
    if not cli.exists():
        subprocess.run(["make", "-C", str(ROOT / "oracle"), "oracle_cli"], capture_output=True)
    return str(cli), "port", "oracle/peaq_oracle.c"


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _run_json(cmd, timeout):
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout)
    if out.returncode != 0:
        raise RuntimeError(f"{cmd[0]} failed: {out.stderr[-300:]}")
    return json.loads(out.stdout.strip().splitlines()[-1].replace('"nan"', "NaN").replace('"inf"', "Infinity")
                      .replace('"-inf"', "-Infinity"))


def cpu_single(n_samples, channels, seed0, advanced, budget_s):
    """One core, bounded sample of the same seeded pairs; also returns the per-pair results."""
    tool, kind, what = _cpu_tool()
    per_pair = {("reference", False): 0.055, ("reference", True): 0.45, ("port", False): 0.3, ("port", True): 1.5}
    pairs = max(2, int(budget_s / per_pair[(kind, bool(advanced))] * (n_samples / 480000.0) ** -1))
    d = _run_json([tool, "time", str(int(advanced)), str(channels), str(seed0), str(pairs), str(n_samples)], 600)
    base = dict(value=d["frame_pairs_per_s"], unit="frame-pairs/s", cores=1, kind=kind,
                sample=f"{d['pairs']} of the same seeded pairs ({d['frame_pairs']} frame-pairs) through {what}, "
                       f"1 thread, {d['seconds']:.1f} s")
    return base, d


def _cpu_limit():
    """(usable cores, note): the affinity mask, cut down to the container's CPU quota if it has one
    (cgroup v2 cpu.max / v1 cfs quota) -- more busy processes than that only time-slice."""
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    if quota is not None and quota < cores:
        return max(1, int(quota)), f"container CPU quota {quota:g} of {cores} visible cores"
    return cores, None


def cpu_all_cores(n_samples, channels, seed0, advanced, budget_s=8.0):
    """One process per usable host core, each pushing its own seeded pairs through the element over and
    over for `budget_s` seconds; the timed regions start together (wall-clock rendezvous);
    value = all frame-pairs / (last end - first begin)."""
    tool, kind, what = _cpu_tool()
    cores, note = _cpu_limit()
    distinct = 2
    start = time.time() + 5.0 + cores * 0.02               # generation (~0.3 s per pair) + process start-up
    procs = [subprocess.Popen([tool, "time", str(int(advanced)), str(channels), str(seed0 + 100000 + distinct * i),
                               str(distinct), str(n_samples), str(-int(budget_s)), f"{start:.3f}"],
                              stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for i in range(cores)]
    res = []
    deadline = start + budget_s + 120.0
    for p in procs:
        try:
            out, _ = p.communicate(timeout=max(1.0, deadline - time.time()))
            if p.returncode == 0:
                res.append(json.loads(out.strip().splitlines()[-1].replace('"nan"', "NaN")))
        except Exception:
            p.kill()
    if not res:
        return None
    frames = sum(r["frame_pairs"] for r in res)
    wall = max(r["t_end"] for r in res) - min(r["t_begin"] for r in res)
    late = sum(1 for r in res if r["t_begin"] > start + 0.25)
    return dict(value=frames / wall, unit="frame-pairs/s", cores=len(res), nproc=os.cpu_count(), cpu_model=_cpu_model(),
                kind=kind,
                sample=f"{len(res)} processes (one per usable core) looping over {distinct} seeded pairs each for "
                       f"{budget_s:g} s = {frames} frame-pairs through {what}, timed regions started together, "
                       f"{wall:.1f} s wall" + (f"; {note}" if note else "")
                       + (f" ({late} processes started late)" if late else ""))


def cpu_records(n_samples, channels, seed0, advanced, pairs, procs=None, timeout=3000):
    """The reference's per-pair results (MOVs, DI, ODG) for the seeded pairs seed0 .. seed0 + pairs - 1, one process
    per usable core, each on its own contiguous block of seeds (`ref_harness time`, repeats = 0: every pair generated,
    then run through the element once).  -> dict(odg, di, movs (flat, 11 per pair), pairs, seconds, cores, kind)."""
    tool, kind, what = _cpu_tool()
    cores, _ = _cpu_limit()
    procs = max(1, min(procs or cores, pairs))
    per, extra = divmod(pairs, procs)
    blocks, s = [], seed0
    for i in range(procs):
        n = per + (1 if i < extra else 0)
        blocks.append((s, n))
        s += n
    t0 = time.time()
    running = [(subprocess.Popen([tool, "time", str(int(advanced)), str(channels), str(b0), str(n), str(n_samples)],
                                 stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True), n) for b0, n in blocks if n]
    out = dict(odg=[], di=[], movs=[])
    for p, n in running:
        o, e = p.communicate(timeout=max(1.0, t0 + timeout - time.time()))
        if p.returncode != 0:
            for q, _ in running:
                q.kill()
            raise RuntimeError(f"{tool} failed: {e[-300:]}")
        d = json.loads(o.strip().splitlines()[-1].replace('"nan"', "NaN").replace('"inf"', "Infinity").replace('"-inf"', "-Infinity"))
        if len(d["odg"]) != n:
            raise RuntimeError(f"{tool}: {len(d['odg'])} records for {n} pairs")
        for k in out:
            out[k] += d[k]
    out.update(pairs=pairs, seconds=time.time() - t0, cores=len(running), kind=kind, what=what)
    return out


# MOVs behind a discrete gate (a threshold decides whether a frame counts at all): the ones where a last-bit
# difference upstream could show as a step instead of as a last-bit difference
GATED_MOVS = ("BandwidthRefB", "BandwidthTestB", "RelDistFramesB", "ADBB", "MFPDB")


def result_deltas(gpu_rows, cpu, advanced):
    """GPU vs the reference's results over the pairs `cpu` holds (same seeds, same order): maxima and 99th percentiles
    of |delta ODG|, |delta DI|, the largest relative difference per MOV, NaN mismatches, and for the discretely gated
    MOVs the number of pairs on which they differ AT ALL beyond rounding (relative 1e-9)."""
    import numpy as np
    from gstpeaq_amd.capi import MOV_NAMES_ADVANCED, MOV_NAMES_BASIC
    n = len(cpu["odg"])
    g = np.asarray(gpu_rows[:n], dtype=np.float64)
    odg_c, di_c = np.asarray(cpu["odg"], dtype=np.float64), np.asarray(cpu["di"], dtype=np.float64)
    movs_c = np.asarray(cpu["movs"], dtype=np.float64).reshape(n, 11)
    names = MOV_NAMES_ADVANCED if advanced else MOV_NAMES_BASIC
    nan_mismatch = int((np.isnan(g[:, 12]) != np.isnan(odg_c)).sum())
    ok = ~np.isnan(odg_c) & ~np.isnan(g[:, 12])
    mov_rel, gated = {}, {}
    for i, name in enumerate(names):
        a, b = g[ok, i], movs_c[ok, i]
        den = np.maximum(np.maximum(np.abs(a), np.abs(b)), 1e-12)
        rel = np.abs(a - b) / den
        mov_rel[name] = float(np.max(rel)) if a.size else None
        if name in GATED_MOVS:
            gated[name] = int((rel > 1e-9).sum())
    d_odg, d_di = np.abs(g[ok, 12] - odg_c[ok]), np.abs(g[ok, 11] - di_c[ok])
    return dict(odg_max_abs_delta=float(np.max(d_odg)) if ok.any() else None,
                odg_p99_abs_delta=float(np.percentile(d_odg, 99)) if ok.any() else None,
                di_max_abs_delta=float(np.max(d_di)) if ok.any() else None,
                di_p99_abs_delta=float(np.percentile(d_di, 99)) if ok.any() else None,
                mov_max_rel_delta=mov_rel, gated_movs_pairs_differing=gated,
                delta_pairs=int(n), delta_nan_mismatches=nan_mismatch)


# ----------------------------------------------------------------------------------------------
def device_power_info(index):
    """Best effort, never fatal: the power cap and what the board reports right now (sysfs hwmon of the amdgpu card,
    else rocm-smi) -- with the calibration kernel's clock what tells a slow box from a slow library."""
    import glob
    info = {}
    try:
        cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device/hwmon/hwmon*"))
        if cards:
            h = cards[min(index, len(cards) - 1)]

            def rd(name, scale):
                try:
                    return float(open(f"{h}/{name}").read().strip()) * scale
                except Exception:
                    return None
            uid = None
            try:
                uid = open(h.split("/hwmon/")[0] + "/unique_id").read().strip()   # which board this line was measured on
            except Exception:
                pass
            info = {"source": h, "unique_id": uid, "power_cap_w": rd("power1_cap", 1e-6), "power_cap_max_w": rd("power1_cap_max", 1e-6),
                    "power_now_w": rd("power1_average", 1e-6) or rd("power1_input", 1e-6),
                    "sclk_now_mhz": rd("freq1_input", 1e-6), "temp_c": rd("temp1_input", 1e-3)}
    except Exception:
        pass
    if not info.get("power_cap_w"):
        try:
            out = subprocess.run(["rocm-smi", "-d", str(index), "--showmaxpower", "--showpower", "--showclocks", "--json"],
                                 capture_output=True, text=True, timeout=20).stdout
            d = json.loads(out[out.index("{"):])
            card = next(iter(d.values()))
            info = {"source": "rocm-smi", "raw": {k: v for k, v in card.items() if any(t in k.lower() for t in ("power", "sclk"))}}
        except Exception as e:
            info = info or {"error": repr(e)[:120]}
    return info


class DeviceSampler:
    """What the board reports WHILE the timed steps run (sysfs hwmon of the amdgpu card, every 10 ms from a thread of
    its own; the main thread sits in a device synchronisation meanwhile): socket power, shader and memory clock.
    Best effort, never fatal -- with the in-kernel step clock what is left to tell two boxes apart."""

    def __init__(self, index):
        import glob
        self.h = None
        try:
            cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device/hwmon/hwmon*"))
            if cards:
                self.h = cards[min(index, len(cards) - 1)]
        except Exception:
            pass
        self.rows, self._stop, self._t = [], False, None

    def _rd(self, name, scale):
        try:
            return float(open(f"{self.h}/{name}").read().strip()) * scale
        except Exception:
            return None

    def _run(self):
        while not self._stop:
            self.rows.append((self._rd("power1_average", 1e-6) or self._rd("power1_input", 1e-6),
                              self._rd("freq1_input", 1e-6), self._rd("freq2_input", 1e-6), self._rd("temp1_input", 1e-3)))
            time.sleep(0.01)

    def __enter__(self):
        if self.h:
            import threading
            self._t = threading.Thread(target=self._run, daemon=True)
            self._t.start()
        return self

    def __exit__(self, *a):
        self._stop = True
        if self._t:
            self._t.join(timeout=1.0)

    def summary(self):
        def col(i):
            v = [r[i] for r in self.rows if r[i] is not None]
            return dict(mean=sum(v) / len(v), min=min(v), max=max(v)) if v else None
        if not self.rows:
            return None
        return dict(samples=len(self.rows), power_w=col(0), sclk_mhz=col(1), mclk_mhz=col(2), temp_c=col(3),
                    source=self.h, how="hwmon read every 10 ms while the timed steps ran")


def source_hash():
    """sha256 over the sources the profiled kernels (basic front and back end) are compiled from, with comments and
    white space taken out: what the counter profile is valid for (a reworded comment does not make it stale, a
    changed statement does; the filter-bank kernels and the host code have no part in it)"""
    import hashlib
    import re
    h = hashlib.sha256()
    names = ("peaq_frontend.hip", "peaq_backend.hip", "peaq_wave.h", "peaq_device.h", "peaq_kernels.h", "Makefile")
    for f in sorted((ROOT / "gstpeaq_amd" / "csrc").glob("*")):
        if f.is_file() and f.name in names:
            text = f.read_text(errors="replace")
            if f.suffix in (".hip", ".h", ".cpp"):
                text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
                text = re.sub(r"//[^\n]*", "", text)
            h.update(f.name.encode())
            h.update("".join(text.split()).encode())
    return h.hexdigest()[:16]


def fb_source_hash():
    """the same for the filter-bank kernels (profiles/pmc_frontend.json, key fb_bank_kernel<MfmaF64>)"""
    import hashlib
    import re
    h = hashlib.sha256()
    for name in ("peaq_device.h", "peaq_fb.hip", "peaq_kernels.h", "peaq_wave.h", "Makefile"):
        text = (ROOT / "gstpeaq_amd" / "csrc" / name).read_text(errors="replace")
        if not name == "Makefile":
            text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
            text = re.sub(r"//[^\n]*", "", text)
        h.update(name.encode())
        h.update("".join(text.split()).encode())
    return h.hexdigest()[:16]


def bank_profile_numbers():
    """Instruction counts of fb_bank_kernel<MfmaF64> per wave and filter-bank block from the committed counter profile
    (NOT measured in this run; stale if the kernel's sources changed since)."""
    prof = ROOT / "profiles" / "pmc_frontend.json"
    try:
        d = json.loads(prof.read_text()).get("fb_bank_kernel<MfmaF64>")
        if not d:
            return None
        now = fb_source_hash()
        return dict(d, from_profile=dict(file="profiles/pmc_frontend.json", commit=d.get("commit"), source_hash=d.get("source_hash"),
                                         source_hash_now=now, stale=d.get("source_hash") != now))
    except Exception:
        return None


def profile_numbers():
    """Counter-derived figures of the front end and the back end.  They are NOT measured in this run: they come
    from separate rocprofv3 --pmc passes of this same command (tools/collect_profiles.sh, tools/pmc_mix.sh),
    committed under profiles/ together with the commit, the hash of the kernel sources and the kernel time they
    were taken at.  A profile taken from other sources than the ones present is marked stale."""
    out = dict(traffic=None, valu_per_wave=None, from_profile=None)
    prof = ROOT / "profiles" / "pmc_frontend.json"
    if prof.exists():
        try:
            d = json.loads(prof.read_text())
            out["traffic"] = d.get("hbm_bytes_per_launch")
            out["valu_per_wave"] = d.get("valu_insts_per_wave")
            out["fp64_per_wave"] = d.get("valu_fp64_insts_per_wave")
            be = d.get("backend_kernel<109,false>", {})
            out["be_valu_per_wave_frame"] = be.get("valu_insts_per_wave_frame")
            out["be_fp64_per_wave_frame"] = be.get("valu_fp64_insts_per_wave_frame")
            now = source_hash()
            out["from_profile"] = dict(file="profiles/pmc_frontend.json", commit=d.get("commit"),
                                       source_hash=d.get("source_hash"), source_hash_now=now,
                                       stale=d.get("source_hash") != now,
                                       kernel_avg_ms_in_profile=d.get("kernel_avg_ms"),
                                       launches_in_profile=d.get("launches"))
        except Exception:
            pass
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--pairs", type=int, default=None,
                    help="pairs per GPU and step (default: 4096 = configs[1] at --gpus 1, 32768 = configs[3] beyond)")
    ap.add_argument("--wave-pairs", type=int, default=WAVE_PAIRS, help="pairs resident at a time (waves mode)")
    ap.add_argument("--waves", action="store_true", help="consume the pairs in waves also at --gpus 1")
    ap.add_argument("--seconds", type=float, default=10.0, help="length of each pair")
    ap.add_argument("--channels", type=int, default=2)
    ap.add_argument("--advanced", action="store_true", help="configs[2] as the main metric instead of configs[1]")
    ap.add_argument("--no-advanced", action="store_true", help="skip the `advanced` sub-object")
    ap.add_argument("--reduced-precision", action="store_true",
                    help="with --advanced: time the opt-in split-FP16 FIR (PEAQ_FIR_F16X3) instead of the default all-FP64 engine")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--delta-pairs", type=int, default=256,
                    help="pairs whose results are compared with the reference's (both versions; all host cores)")
    ap.add_argument("--no-scaling-reference", action="store_true",
                    help="skip the waves-mode pass that makes an N = 1 line comparable with N > 1 lines")
    args = ap.parse_args()

    # `python bench.py --gpus N` by itself (no launcher, N > 1): become the launcher -- one process per GPU under
    # torch.distributed.run on 127.0.0.1 with a free port -- so that the driver's N = 1 command shape also works for
    # N = 2, 4, 8.  Under torch.distributed.run (WORLD_SIZE set) nothing changes.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import socket
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        os.environ.setdefault("OMP_NUM_THREADS", "1")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
        if os.environ.get("PEAQ_BENCH_LAUNCH_DRYRUN") == "1":   # tests/test_capi_host.py: the command, not the run
            print(json.dumps({"launch": cmd}))
            return
        sys.stdout.flush()
        os.execv(sys.executable, cmd)

    import numpy as np
    import torch
    import gstpeaq_amd
    from gstpeaq_amd import parallel

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} inside a job of WORLD_SIZE {world}: launch it as `python bench.py "
                         f"--gpus N` (it starts its own ranks) or under torch.distributed.run --nproc-per-node N")
    # PEAQ_BENCH_DIST_BACKEND=gloo: the N > 1 path with a CPU communicator, every rank on the GPU(s) the box has
    # (RCCL refuses two ranks on one device) -- how tests/test_gpu_two_ranks.py runs two REAL ranks on a one-GPU
    # box: shards, seeds, waves, timing reduction and the gather (staged through host memory) are the production
    # code, only the transport differs.  Default: "nccl" = RCCL over xGMI.
    backend = os.environ.get("PEAQ_BENCH_DIST_BACKEND", "nccl")
    if backend == "gloo":
        local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    # PEAQ_BENCH_FORCE_DIST=1: build the communicator also for one rank, so that a one-GPU box can run the
    # RCCL calls of the N > 1 path (tests/test_gpu_rccl_single.py)
    if world > 1 or os.environ.get("PEAQ_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)  # "nccl" is RCCL on ROCm
    cdev = torch.device("cpu") if backend == "gloo" else dev   # where the collectives' small tensors live

    waves_mode = args.waves or world > 1
    pairs_per_gpu = args.pairs or (CONFIG4_PAIRS_PER_GPU if waves_mode else 4096)
    wave_pairs = min(args.wave_pairs, pairs_per_gpu) if waves_mode else pairs_per_gpu
    total_pairs = pairs_per_gpu * world
    lo, hi = parallel.shard(total_pairs, rank, world)       # this rank's block of pair indices
    n_samples = int(round(args.seconds * 48000))
    seed_base = 1                                            # pair i of the job carries seed seed_base + i

    ctx = gstpeaq_amd.Context(local_rank)
    assert ctx.fir_mode() == "f64" or os.environ.get("PEAQ_AMD_FIR") or os.environ.get("PEAQ_AMD_FIR_FP64"), \
        "the engine's default must be the reference's FP64 arithmetic"
    if args.advanced:                # configs[2] as the main metric: the default engine unless asked otherwise
        ctx.set_fir_mode("f16x3" if args.reduced_precision else "f64")
    ref, test = gstpeaq_amd.synth_fill(ctx, seed_base + lo, wave_pairs, args.channels, n_samples, device=dev)
    results = torch.empty((hi - lo, 16), dtype=torch.float64, device=dev)
    torch.cuda.synchronize(dev)

    def barrier():
        torch.cuda.synchronize(dev)
        if dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    sampled = {}                                           # advanced? -> what the board reported during the timed steps

    def run_steps(advanced, steps, warmup):
        """-> (seconds inside the timed regions, wall seconds, HIP-event timing of the last pass)"""
        if not waves_mode:
            for _ in range(warmup):
                gstpeaq_amd.batch_run(ctx, advanced, ref, test, results=results, sync=False)
            barrier()
            with DeviceSampler(local_rank) as smp:
                t0 = time.perf_counter()
                for _ in range(steps):
                    gstpeaq_amd.batch_run(ctx, advanced, ref, test, results=results, sync=False)
                barrier()
                el = time.perf_counter() - t0
            sampled[bool(advanced)] = smp.summary()
            return el, el, ctx.last_timing()
        # waves: the resident buffers hold wave 0 on entry; a warm-up step runs that wave only
        for _ in range(warmup):
            gstpeaq_amd.batch_run(ctx, advanced, ref, test, results=results[:wave_pairs], sync=False)
        barrier()
        w0 = time.perf_counter()
        timed = 0.0
        for _ in range(steps):
            timed += parallel.run_waves(ctx, advanced, seed_base + lo, hi - lo, wave_pairs, ref, test, results)
        barrier()
        wall = time.perf_counter() - w0
        gstpeaq_amd.synth_fill(ctx, seed_base + lo, wave_pairs, args.channels, n_samples, out=(ref, test))
        return timed, wall, ctx.last_timing()

    def reduce_max(x):
        t = torch.tensor([x], dtype=torch.float64, device=cdev)
        if dist:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def measure(advanced, steps, warmup):
        timed, wall, timing = run_steps(advanced, steps, warmup)
        timed, wall = reduce_max(timed), reduce_max(wall)
        # the one collective of the path: gather the per-pair result records; its own time
        torch.cuda.synchronize(dev)
        tg = time.perf_counter()
        gathered = parallel.gather_results(results, world, dist)
        torch.cuda.synchronize(dev)
        gather_ms = (time.perf_counter() - tg) * 1e3
        if rank == 0 and os.environ.get("PEAQ_BENCH_DUMP_RESULTS"):   # tests: the job's result records in pair order
            import numpy as np
            np.save(os.environ["PEAQ_BENCH_DUMP_RESULTS"], gathered.cpu().numpy())
        frame_pairs_all = float(gathered[:, 14].sum().item())
        frame_pairs_rank = float(results[:, 14].sum().item())
        return dict(timed=timed, wall=wall, timing=timing, gathered=gathered, fp_all=frame_pairs_all, gather_ms=gather_ms,
                    fp_rank=frame_pairs_rank, rows=results[: min(4096, hi - lo)].cpu().numpy().copy())

    def simd_cycles_per_s(clock_mhz):
        """SIMDs x the shader clock THIS run measured (compute units from peaq_calibrate, the clock of the step itself);
        the constants 1024 x 2.4 GHz only if neither is known"""
        cus = (cal0 or {}).get("compute_units") or 256
        return cus * 4 * (clock_mhz * 1e6 if clock_mhz else 2.4e9)

    def frontend_roofline(m, advanced, clock_mhz=None):
        timing = m["timing"]
        # HIP events of the LAST batch call (one wave in waves mode) on the launch stream
        last_pairs = wave_pairs if waves_mode else pairs_per_gpu
        fp_last = m["fp_rank"] * last_pairs / (hi - lo)
        if waves_mode and (hi - lo) % wave_pairs:
            fp_last = m["fp_rank"] / (hi - lo) * ((hi - lo) % wave_pairs)
        fe_s = timing["frontend_ms"] * 1e-3
        achieved = fp_last * ALGO_BYTES_PER_FRAME_PAIR / fe_s / 1e9 if fe_s > 0 else 0.0
        prof = profile_numbers()
        valu_frac = step_valu_frac = None
        stale = bool((prof.get("from_profile") or {}).get("stale"))
        if prof.get("valu_per_wave") and not advanced and fe_s > 0 and not stale:   # (a stale profile prices nothing)
            # One issue model everywhere: an FP64 vector instruction occupies a SIMD for 4 cycles
            # per wave, every other one for 2 (MI355X_MICROARCH.md: SIMD-32, FP64 at half the FP32 rate;
            # profiles/r02_microbench.txt measures 4.5-5.4 and 2.4-2.9 for back-to-back FMAs); the device's SIMDs at
            # the shader clock the step itself measured (simd_cycles_per_s).
            # Instruction counts per wave from the committed counter profile, times from THIS run's HIP events.
            f64 = prof.get("fp64_per_wave") or 0.0
            cyc = f64 * 4 + (prof["valu_per_wave"] - f64) * 2
            fe_issue_s = fp_last * args.channels * 2 * cyc / simd_cycles_per_s(clock_mhz)
            valu_frac = fe_issue_s / fe_s
            if prof.get("be_valu_per_wave_frame") and timing["total_ms"] > 0:
                b64 = prof.get("be_fp64_per_wave_frame") or 0.0
                be_cyc = b64 * 4 + (prof["be_valu_per_wave_frame"] - b64) * 2
                be_issue_s = fp_last * args.channels * be_cyc / simd_cycles_per_s(clock_mhz)
                step_valu_frac = (fe_issue_s + be_issue_s) / (timing["total_ms"] * 1e-3)
        launches = max(timing["frontend_launches"], 1)
        return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": prof["traffic"],
                "kernel": "frontend_kernel<55>" if advanced else "frontend_kernel<109>", "launches": launches,
                "avg_launch_ms": timing["frontend_ms"] / launches,
                "algorithmic_bytes_per_launch": fp_last * ALGO_BYTES_PER_FRAME_PAIR / launches,
                "compute_frac_fp64_vector": fp_last / fe_s * FLOP_PER_FRAME_PAIR / (FP64_VECTOR_PEAK_TFLOPS * 1e12)
                if fe_s > 0 else None,
                "valu_issue_frac": valu_frac, "step_valu_issue_frac": step_valu_frac,
                "valu_issue_clock_mhz": clock_mhz,
                "valu_issue_model": "4 cycles per FP64 vector instruction and wave, 2 per other vector instruction, SIMDs x the "
                                    "step's own shader clock (null if the counter profile is stale); "
                                    "kernel: front-end instructions / front-end kernel time; step: front + back end / "
                                    "whole step (HIP events of this run; instruction counts per wave from_profile)",
                "from_profile": prof["from_profile"],
                "from_profile_fields": ["traffic", "valu_issue_frac", "step_valu_issue_frac"],
                "backend_ms": timing["backend_ms"], "fb_ms": timing["fb_ms"], "step_ms_events": timing["total_ms"]}

    def filterbank_roofline(m, clock_mhz=None):
        # configs[2]: the dominant kernel is the filter bank (fb_bank_kernel), a folded FIR bank on the
        # matrix cores.  Algorithmic work as the reference counts it (fbearmodel.c:404-434): per tap
        # pair two additions and two multiply-adds, 10 914 tap pairs per sub-sample, 6 sub-samples per
        # 192-sample block; against the FP64 matrix peak (= FP64 vector peak, 78.6 TFLOP/s).
        timing = m["timing"]
        last_pairs = wave_pairs if waves_mode else pairs_per_gpu
        blocks = float(results[:last_pairs, 15].sum().item())
        # 21 828 multiply-adds per sub-sample and signal (10 914 folded tap pairs x {re, im}) = the
        # ALGORITHMIC work `achieved` is quoted in; the reference's own loop spends 65 484 flop on them (two additions
        # and two multiply-adds per tap pair, fbearmodel.c:404-434): `reference_operation_frac`
        flops = blocks * args.channels * 2 * 6 * 21828 * 2
        ref_flops = blocks * args.channels * 2 * 6 * 10914 * 6
        fb_s = timing["fb_ms"] * 1e-3
        tf = flops / fb_s / 1e12 if fb_s > 0 else 0.0
        mode = ctx.fir_mode()
        peak = {"f64": FP64_VECTOR_PEAK_TFLOPS, "f32": FP32_MATRIX_PEAK_TFLOPS, "f16x3": FP16_MATRIX_PEAK_TFLOPS}[mode]
        extra = {}
        if mode == "f64":
            # The FP64 engine does not evaluate the reference's sums tap by tap any more ("block-sum
            # form"): per sub-sample and signal its matrix instructions carry 24 bands x 8 rows x 32 multiply-adds
            # (enter rows of the three rectangular windows + the block the window ends in) and one direct tile of
            # 30 K steps x 4 delays x 16 bands x {re, im} for the 16 short filters.  `achieved` / `frac` stay what
            # they were -- the REFERENCE's operation count over this kernel's time, comparable from round to round
            # (it may pass 1: the kernel does less than it is credited with) -- and the kernel's own count stands
            # beside them: the fraction of the matrix pipe's time its matrix instructions fill.
            issued = blocks * args.channels * 2 * 6 * (24 * 8 * 32 + 30 * 4 * 16 * 2) * 2
            # what the kernel's SIMDs are busy with: vector and matrix instructions exclude each other on a SIMD,
            # so issue cycles add -- 4 per FP64 vector instruction and wave, 2 per other vector instruction, 64 per
            # v_mfma_f64_16x16x4_f64; instruction counts per wave and block from the committed counter profile, the
            # kernel's time from THIS run's HIP events; 1024 SIMDs at 2.4 GHz
            bp = bank_profile_numbers()
            simd_busy = None
            if bp and fb_s > 0 and not bp["from_profile"]["stale"]:      # (a stale profile prices nothing)
                cyc = bp["valu_fp64_per_wave_block"] * 4 + (bp["valu_per_wave_block"] - bp["valu_fp64_per_wave_block"]) * 2 \
                    + bp["mfma_per_wave_block"] * 64
                simd_busy = blocks * args.channels * 2 * 4 * cyc / simd_cycles_per_s(clock_mhz) / fb_s
            extra = {"simd_busy_frac": simd_busy, "simd_busy_from_profile": bp and bp["from_profile"],
                     "simd_busy_clock_mhz": clock_mhz,
                     "simd_busy_is": "(4 x FP64 vector + 2 x other vector + 64 x matrix instructions per wave) x waves / (SIMDs x "
                                     "the step's own shader clock x kernel time); instruction counts from the committed counter "
                                     "profile, null if that profile is stale",
                     "reference_flop_per_subsample": 10914 * 6, "algorithmic_flop_per_subsample": 21828 * 2,
                     "matrix_flop_per_subsample": (24 * 8 * 32 + 30 * 4 * 16 * 2) * 2,
                     "matrix_flop_per_launch": issued / max(timing["fb_launches"], 1),
                     "matrix_pipe_frac": issued / fb_s / 1e12 / peak if fb_s > 0 else None,
                     "algorithm": "block-sum form: bands 0..23 as running sums over 32-sample blocks (Hann window = three "
                                  "rectangular windows), bands 24..39 direct; fbearmodel.c:399-435"}
        # `frac` = achieved / peak in every engine, with `achieved` = SURVEY.md 8(d)'s algorithmic multiply-adds over
        # this kernel's time (what the block-sum form is credited with: it issues fewer); what the SIMDs are busy
        # with stands beside it as `simd_busy_frac`, the matrix pipe's own share as `matrix_pipe_frac`
        return {"bound": "mfma", "achieved": tf, "peak": peak, "unit": "TFLOP/s",
                "frac": tf / peak, "frac_is": "achieved / peak; achieved = 21 828 multiply-adds per sub-sample and signal "
                                              "(SURVEY.md 8(d)) x 2 / kernel time",
                "reference_operation_frac": ref_flops / fb_s / 1e12 / peak if fb_s > 0 else None,
                "traffic": None, **extra,
                "kernel": {"f64": "fb_bank_kernel<MfmaF64>", "f32": "fb_bank_kernel<MfmaF32>", "f16x3": "fb_bank_kernel<MfmaH3>"}[mode],
                "peak_is": {"f64": "FP64 matrix = vector peak", "f32": "FP32 matrix peak (f32-input MFMA), MI355X_MICROARCH.md",
                            "f16x3": "dense FP16 matrix peak, MI355X_MICROARCH.md; the kernel issues 6x the algorithmic "
                                     "multiply-adds (three FP16 products per term, the fold of the symmetric taps undone) "
                                     "and spends most of its time in its FP64 vector phases, DESIGN.md 3"}[mode],
                "launches": timing["fb_launches"], "avg_launch_ms": timing["fb_ms"] / max(timing["fb_launches"], 1),
                "algorithmic_flop_per_launch": flops / max(timing["fb_launches"], 1),
                "frontend_ms": timing["frontend_ms"], "backend_ms": timing["backend_ms"],
                "fb_ms": timing["fb_ms"], "step_ms_events": timing["total_ms"]}

    main_adv = bool(args.advanced)
    # the device's clock under a fixed FP64 load before and after the timed region, and (below) while the step itself ran
    cal0 = ctx.calibrate() if rank == 0 else None
    m = measure(main_adv, args.steps, args.warmup)
    step_clock_mhz = ctx.last_clock_mhz()
    cal1 = ctx.calibrate() if rank == 0 else None
    odg = m["gathered"][:, 12]
    line = None
    if rank == 0:
        value = m["fp_all"] * args.steps / m["timed"]
        if waves_mode:
            workload = (f"{'Advanced' if main_adv else 'Basic'} PEAQ, {total_pairs} synthetic 48 kHz "
                        f"{'stereo' if args.channels == 2 else 'mono'} {args.seconds:g} s (ref,test) pairs sharded "
                        f"over {world} GPU(s), {pairs_per_gpu} per GPU, consumed in waves of {wave_pairs} "
                        f"(generated on the device between the timed regions) -- BASELINE.json configs[3]'s sharding")
        else:
            workload = (f"{'Advanced' if main_adv else 'Basic'} PEAQ, {pairs_per_gpu} synthetic 48 kHz "
                        f"{'stereo' if args.channels == 2 else 'mono'} {args.seconds:g} s (ref,test) pairs per GPU, "
                        f"inputs resident in HBM (BASELINE.json configs[{2 if main_adv else 1}])")
        line = {
            "metric": "2048-sample stereo ref/test frame-pairs/sec (basic PEAQ, ear model -> MOVs -> ODG); ODG delta vs CPU ref"
                      if not main_adv else "FFT frame-pairs/sec (advanced PEAQ incl. filter-bank blocks); ODG delta vs CPU ref",
            "value": value,
            "unit": "frame-pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": m["timed"] / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": workload, "pairs_per_gpu": pairs_per_gpu, "total_pairs": total_pairs,
                       "frame_pairs_per_pair": m["fp_rank"] / (hi - lo),
                       "waves_per_step": (hi - lo + wave_pairs - 1) // wave_pairs if waves_mode else 1,
                       "parallelism": f"pairs sharded over {world} GPU(s), no data-path collective",
                       "result_gather": f"{dist.get_backend()} all_gather over {dist.get_world_size()} rank(s)" if dist
                       else "none (one process)"},
            "roofline": filterbank_roofline(m, step_clock_mhz) if main_adv else frontend_roofline(m, False, step_clock_mhz),
            "odg_mean": float(odg[~torch.isnan(odg)].mean().item()),
            "odg_nan": int(torch.isnan(odg).sum().item()),
        }
        line["per_gpu_value"] = value / world
        nominal = (cal0 or {}).get("max_clock_mhz") or 2400.0
        line["device_clock"] = {
            "step_shader_clock_mhz": step_clock_mhz,
            "calibration_before": cal0, "calibration_after": cal1, "nominal_mhz": nominal,
            "power": device_power_info(local_rank), "board_during_steps": sampled.get(main_adv),
            "how": "step clock: workgroup 0 of every back-end launch of the last timed pass reads the shader-clock and the "
                   "constant-rate counter around its lifetime, beside the front end of the next chunk (peaq_batch_last_clock); "
                   "calibration: peaq_calibrate, a fixed v_fma_f64 kernel (one wave per SIMD, sixteen chains, ~70 ms, alone on the device) "
                   "before and after the timed region -- its second half is the steady state (shader_clock_mhz, fp64_tflops, "
                   "cycles_per_fma: 4 = the FP64 pipe), its first half the ramp from the idle clock "
                   "(ramp_*); rank 0's device"}
        # the same line at the device's nominal clock: the path is issue-bound, its time scales with 1 / clock
        line["value_at_nominal_clock"] = value * nominal / step_clock_mhz if step_clock_mhz else None
        line["result_gather_ms"] = m["gather_ms"]           # after the timed region; 128 B per pair
        if waves_mode:
            line["wall_ms_per_step_incl_generation"] = m["wall"] / args.steps * 1e3
            line["scaling_note"] = ("every rank runs the waves regime; compare per_gpu_value with "
                                    "`scaling_reference.value` of the N = 1 line (same regime on one GPU), not with its `value`")
    rows_main = m["rows"]

    # ---- configs[2] next to configs[1] in the same line ---------------------------------------------
    # `advanced.value` is the engine's default: the REFERENCE's precision (every stage FP64, PEAQ_FIR_F64: the mode
    # the 1e-9 / 1e-7 parity tests hold), with the roofline of ITS bank kernel from this run's HIP events; the opt-in
    # split-FP16 FIR + FP32 slopes / spreading is the labelled sub-object `reduced_precision_f16x3` with its own
    # roofline and its deviation from the default on these pairs.
    adv = None
    rows_adv = None
    if not main_adv and not args.no_advanced:
        adv_steps = max(1, min(args.steps, 3))
        default_mode = ctx.fir_mode()
        wl = line["config"]["workload"].replace("Basic PEAQ", "Advanced PEAQ").replace("configs[1]", "configs[2]") if rank == 0 else ""
        try:
            ctx.set_fir_mode("f64")
            m64 = measure(True, adv_steps, 1)
            adv_clock = ctx.last_clock_mhz()
            if rank == 0:
                adv = {"config": wl, "step_shader_clock_mhz": adv_clock,
                       "metric": "FFT frame-pairs/sec (advanced PEAQ: 55-band FFT model + 40-band filter bank, 5 MOVs)",
                       "value": m64["fp_all"] * adv_steps / m64["timed"], "unit": "frame-pairs/s", "steps": adv_steps,
                       "warmup": 1, "ms_per_step": m64["timed"] / adv_steps * 1e3,
                       "fb_blocks_per_frame_pair": float(m64["gathered"][:, 15].sum().item()) / max(m64["fp_all"], 1.0),
                       "dtype": "f64 (every stage, like the reference; FIR bank of the filter-bank ear model on v_mfma_f64_16x16x4_f64)",
                       "engine": "PEAQ_FIR_F64: the engine's default (peaq_ctx_create); the opt-in fast engine is reduced_precision_f16x3",
                       "roofline": filterbank_roofline(m64, adv_clock), "board_during_steps": sampled.get(True),
                       "odg_mean": float(m64["gathered"][:, 12][~torch.isnan(m64["gathered"][:, 12])].mean().item())}
            rows_adv = m64["rows"]
        except Exception as e:                               # the bench line must survive this leg
            if rank == 0:
                adv = {"error": repr(e)[:300]}
        finally:
            ctx.set_fir_mode(default_mode)
        if True:
            try:
                ctx.set_fir_mode("f16x3")
                ma = measure(True, adv_steps, 1)
                if rank == 0 and adv is not None:
                    sub = {"value": ma["fp_all"] * adv_steps / ma["timed"], "unit": "frame-pairs/s", "steps": adv_steps,
                           "ms_per_step": ma["timed"] / adv_steps * 1e3,
                           "dtype": "FIR bank on v_mfma_f32_16x16x32_f16 with both operands split into two FP16 parts (three "
                                    "products per term, FP32 accumulation), slopes and upward spreading FP32, everything else "
                                    "FP64: narrower than the reference's arithmetic, hence opt-in and not the headline",
                           "engine": "PEAQ_FIR_F16X3 (peaq_ctx_set_fir_mode / PEAQ_AMD_FIR=f16x3); NOT the default",
                           "roofline": filterbank_roofline(ma, ctx.last_clock_mhz())}
                    if rows_adv is not None:
                        a, b = ma["rows"][:, 12], rows_adv[:, 12]
                        ok = ~(np.isnan(a) | np.isnan(b))
                        sub["odg_max_abs_delta_vs_fp64_engine"] = float(np.max(np.abs(a[ok] - b[ok]))) if ok.any() else None
                        sub["pairs"] = int(ok.sum())
                    sub["speedup_over_default"] = sub["value"] / adv["value"] if adv.get("value") else None
                    adv["reduced_precision_f16x3"] = sub
            except Exception as e:
                if rank == 0 and adv is not None:
                    adv["reduced_precision_f16x3"] = {"error": repr(e)[:300]}
            finally:
                ctx.set_fir_mode(default_mode)

    # ---- the scaling regime on this GPU (N = 1 only): what every rank of an N > 1 run does -- 32 768 pairs
    # consumed in synchronised waves of 4096 -- so that an efficiency computed from the driver's N = 1, 2, 4, 8
    # lines can compare like with like (`scaling_reference.value`, not `value`, is the per-GPU rate of that regime)
    if world == 1 and not waves_mode and not main_adv and not args.no_scaling_reference and args.pairs is None \
            and rank == 0:
        try:
            wr = torch.empty((CONFIG4_PAIRS_PER_GPU, 16), dtype=torch.float64, device=dev)
            barrier()
            timed_w = parallel.run_waves(ctx, False, seed_base, CONFIG4_PAIRS_PER_GPU, pairs_per_gpu, ref, test, wr)
            barrier()
            line["scaling_reference"] = {
                "value": float(wr[:, 14].sum().item()) / timed_w, "unit": "frame-pairs/s",
                "mode": f"waves: {CONFIG4_PAIRS_PER_GPU} pairs on this GPU in {CONFIG4_PAIRS_PER_GPU // pairs_per_gpu} "
                        f"device-synchronised waves of {pairs_per_gpu} (generation between the timed regions), exactly "
                        "what every rank of `--gpus N`, N > 1, runs; `value` above is the pipelined single-batch regime",
                "ms_per_wave": timed_w / (CONFIG4_PAIRS_PER_GPU // pairs_per_gpu) * 1e3}
            del wr
            gstpeaq_amd.synth_fill(ctx, seed_base + lo, wave_pairs, args.channels, n_samples, out=(ref, test))
        except Exception as e:
            line["scaling_reference"] = {"error": repr(e)[:300]}

    # ---- CPU legs: rank 0.  At N = 1 one core, all usable cores, and the advanced version's leg; at N > 1 the
    # one-core leg only (the other ranks wait at the final barrier meanwhile; nothing is being timed any more).
    # The comparison with the reference's results (`odg_max_abs_delta` ...) runs on the FIRST `--delta-pairs` pairs of
    # the job (256: SURVEY.md 8(d)) in both versions, through one reference process per usable host core; the one-core
    # leg only times.
    if rank == 0 and not args.no_cpu_baseline:
        try:
            base, cpu = cpu_single(n_samples, args.channels, seed_base + lo, main_adv, 8.0)
            line["cpu_baseline"] = base
            n_delta = min(args.delta_pairs, len(rows_main)) if world == 1 else len(cpu["odg"])
            if n_delta > len(cpu["odg"]):
                cpu = cpu_records(n_samples, args.channels, seed_base + lo, main_adv, n_delta)
                line["delta_how"] = f"{cpu['pairs']} pairs through {cpu['what']} on {cpu['cores']} processes, {cpu['seconds']:.1f} s"
            line.update(result_deltas(rows_main, cpu, main_adv))
            line["delta_vs"] = base["kind"]
            if world == 1:
                allc = cpu_all_cores(n_samples, args.channels, seed_base + lo, main_adv)
                if allc:
                    line["cpu_baseline"]["all_cores"] = allc
                if adv is not None and rows_adv is not None:
                    abase, acpu = cpu_single(n_samples, args.channels, seed_base + lo, True, 6.0)
                    adv["cpu_baseline"] = abase
                    n_delta = min(args.delta_pairs, len(rows_adv))
                    if n_delta > len(acpu["odg"]):
                        acpu = cpu_records(n_samples, args.channels, seed_base + lo, True, n_delta)
                        adv["delta_how"] = (f"{acpu['pairs']} pairs through {acpu['what']} on {acpu['cores']} processes, "
                                            f"{acpu['seconds']:.1f} s")
                    adv.update(result_deltas(rows_adv, acpu, True))
                    adv["delta_vs"] = abase["kind"]
        except Exception as e:                               # the bench line must survive a broken baseline leg
            line["cpu_baseline_error"] = repr(e)[:300]
    if rank == 0:
        if adv is not None:
            line["advanced"] = adv
        print(json.dumps(line))
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
