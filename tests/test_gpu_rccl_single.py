"""The N > 1 path of bench.py on the one GPU of the box: `torch.distributed.run` with one rank and
PEAQ_BENCH_FORCE_DIST=1 builds a real RCCL communicator ("nccl" backend) and runs the path's
collectives on it -- barrier, all_reduce(MAX) of the timings, the two all_gathers of
parallel.gather_results -- around a small share consumed in waves.  What it cannot show is more than
one rank (RCCL refuses two ranks on one device); that stays with the driver's 8-GPU run and the
world_size-2 gloo tests.  Needs an MI355X (`-m gpu`)."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def test_bench_collectives_on_a_one_rank_rccl_communicator():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU (there is no CPU fallback in the product)")
    env = dict(os.environ, PEAQ_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
           "--master-addr", "127.0.0.1", "--master-port", "29517", str(ROOT / "bench.py"), "--gpus", "1",
           "--steps", "1", "--warmup", "1", "--waves", "--pairs", "96", "--wave-pairs", "64", "--seconds", "2",
           "--no-advanced", "--no-cpu-baseline"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["config"]["result_gather"] == "nccl all_gather over 1 rank(s)"
    assert line["config"]["total_pairs"] == 96 and line["config"]["waves_per_step"] == 2
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["odg_nan"] == 0
    # 96 pairs of 2 s: (96000 - 2048) // 1024 + 1 full frames + the flush frame = 93 frame pairs each
    assert line["config"]["frame_pairs_per_pair"] == 93
