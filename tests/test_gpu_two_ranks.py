"""The N > 1 path of bench.py with TWO real ranks on the box's one GPU.  RCCL refuses two ranks on one device, so
the communicator is gloo (PEAQ_BENCH_DIST_BACKEND=gloo, bench.py): shards, per-rank seed offsets, the waves of each
rank, the all_reduce(MAX) of the timings and the gather of the HIP-produced result records -- staged through host
memory -- are the production code, only the transport of the collectives differs from the 8-GPU run.  The gathered
records must equal, bit for bit and in pair order, those of ONE process running the same 192 seeds.
Needs an MI355X (`-m gpu`)."""
import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
COMMON = ["--steps", "1", "--warmup", "1", "--wave-pairs", "64", "--seconds", "2", "--no-advanced"]


def run(cmd, env):
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    return json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])


def test_two_ranks_on_one_gpu_equal_one_process(tmp_path):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU (there is no CPU fallback in the product)")
    two, one = tmp_path / "two.npy", tmp_path / "one.npy"
    env2 = dict(os.environ, PEAQ_BENCH_DIST_BACKEND="gloo", PEAQ_BENCH_DUMP_RESULTS=str(two),
                HSA_ENABLE_IPC_MODE_LEGACY="0")
    line2 = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                 "--master-addr", "127.0.0.1", "--master-port", "29531", str(ROOT / "bench.py"), "--gpus", "2",
                 "--pairs", "96"] + COMMON, env2)
    env1 = dict(os.environ, PEAQ_BENCH_DUMP_RESULTS=str(one))
    line1 = run([sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--waves", "--pairs", "192",
                 "--no-cpu-baseline", "--no-scaling-reference"] + COMMON, env1)
    a, b = np.load(two), np.load(one)
    assert a.shape == b.shape == (192, 16)
    assert np.array_equal(a.view(np.uint64), b.view(np.uint64)), "gathered records differ from the one-process run"
    # pair i carries seed 1 + i: its ODG differs from pair to pair, so equal arrays also mean equal ORDER; and the
    # second rank's block really came from the other process
    assert len(np.unique(a[:, 12])) > 150
    # the N = 2 line
    cfg = line2["config"]
    assert line2["n_gpus"] == 2 and cfg["total_pairs"] == 192 and cfg["pairs_per_gpu"] == 96
    assert cfg["result_gather"] == "gloo all_gather over 2 rank(s)" and cfg["waves_per_step"] == 2
    assert line2["scaling"] == "weak" and line2["odg_nan"] == 0
    assert abs(line2["per_gpu_value"] * 2 - line2["value"]) < 1e-6 * line2["value"]
    assert "scaling_note" in line2 and line2["result_gather_ms"] >= 0
    # the one-core CPU baseline also at N > 1 (rank 0, after the timed region)
    cb = line2["cpu_baseline"]
    assert cb["cores"] == 1 and cb["value"] > 0 and cb["kind"] in ("reference", "port")
    # 2 s pairs: 93 frame pairs each, both lines
    assert cfg["frame_pairs_per_pair"] == line1["config"]["frame_pairs_per_pair"] == 93
    assert line1["odg_mean"] == pytest.approx(line2["odg_mean"], abs=1e-12)


def test_bench_gpus_2_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with no launcher around it (the shape of the driver's N = 1 command) starts its own
    two ranks under torch.distributed.run and prints the N = 2 line; same records as the torchrun-launched job."""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU (there is no CPU fallback in the product)")
    dump = tmp_path / "self.npy"
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(PEAQ_BENCH_DIST_BACKEND="gloo", PEAQ_BENCH_DUMP_RESULTS=str(dump), HSA_ENABLE_IPC_MODE_LEGACY="0")
    line = run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--pairs", "96", "--no-cpu-baseline"] + COMMON, env)
    assert line["n_gpus"] == 2 and line["config"]["total_pairs"] == 192
    assert line["config"]["result_gather"] == "gloo all_gather over 2 rank(s)"
    a = np.load(dump)
    assert a.shape == (192, 16) and len(np.unique(a[:, 12])) > 150 and line["odg_nan"] == 0


def test_two_ranks_advanced_version_equal_one_process(tmp_path):
    """The same for configs[2]'s arithmetic: `--advanced` (the filter-bank path's three streams, its chunked rows and
    records per rank) on two real ranks sharing the box's GPU; the gathered records equal the one-process run's bit for
    bit (the default FP64 engine is bit-reproducible) -- and the line carries the advanced roofline."""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU (there is no CPU fallback in the product)")
    two, one = tmp_path / "two_adv.npy", tmp_path / "one_adv.npy"
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env2 = dict(env, PEAQ_BENCH_DIST_BACKEND="gloo", PEAQ_BENCH_DUMP_RESULTS=str(two), HSA_ENABLE_IPC_MODE_LEGACY="0")
    common = ["--steps", "1", "--warmup", "1", "--seconds", "2", "--wave-pairs", "32", "--no-cpu-baseline"]
    # (64 pairs per rank in waves of 32, 128 in waves of 32 in the one process: every launch holds 32 pairs in both,
    # so the filter bank's chunks of blocks are cut alike -- what "bit for bit" needs across differently cut streams
    # is stated in tests/gpu_common.py, "chunks")
    line2 = run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--advanced", "--pairs", "64"] + common, env2)
    env1 = dict(env, PEAQ_BENCH_DUMP_RESULTS=str(one))
    run([sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--advanced", "--waves", "--pairs", "128",
         "--no-scaling-reference"] + common, env1)
    a, b = np.load(two), np.load(one)
    assert a.shape == b.shape == (128, 16)
    assert np.array_equal(a.view(np.uint64), b.view(np.uint64)), "gathered advanced records differ from the one-process run"
    assert len(np.unique(a[:, 12])) > 100 and (a[:, 15] > 0).all()        # filter-bank blocks were counted for every pair
    assert line2["n_gpus"] == 2 and line2["config"]["total_pairs"] == 128 and line2["config"]["waves_per_step"] == 2
    assert line2["roofline"]["kernel"] == "fb_bank_kernel<MfmaF64>" and line2["roofline"]["bound"] == "mfma"
    assert line2["roofline"]["frac"] == pytest.approx(line2["roofline"]["achieved"] / line2["roofline"]["peak"])
