"""The N>1 path on CPU: two processes over gloo shard the pairs, each computes its
shard (here with the CPU oracle, standing in for the per-GPU batch call), and the
gather puts the result records back in pair order -- identical to one process."""
import os
import sys
from pathlib import Path

import numpy as np
import pytest

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
sys.path.insert(0, str(HERE.parent))


def _odg_record(seed):
    import oracle_lib as orc
    import synth_np
    r, t = synth_np.pair(seed, 1, 12000)
    res = orc.run_pair(0, r, t)
    return [float(seed), res["odg"], res["di"], float(res["frames"])]


def _worker(rank, world, n_total, port, out_dir):
    import torch
    import torch.distributed as dist
    from gstpeaq_amd import parallel
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = parallel.shard(n_total, rank, world)
    local = torch.tensor([_odg_record(100 + i) for i in range(lo, hi)], dtype=torch.float64).reshape(hi - lo, 4)
    allr = parallel.gather_results(local, world, dist)
    np.save(Path(out_dir) / f"rank{rank}.npy", allr.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_shard_is_a_partition():
    from gstpeaq_amd import parallel
    for n in (0, 1, 7, 8, 4096, 262144):
        for w in (1, 2, 3, 8):
            spans = [parallel.shard(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.parametrize("n_total", [5, 6])
def test_two_ranks_gather_in_pair_order(tmp_path, n_total):
    import torch.multiprocessing as mp
    port = 29500 + (os.getpid() % 2000) + n_total
    mp.spawn(_worker, args=(2, n_total, port, str(tmp_path)), nprocs=2, join=True)
    single = np.array([_odg_record(100 + i) for i in range(n_total)])
    for r in range(2):
        got = np.load(tmp_path / f"rank{r}.npy")
        assert got.shape == single.shape
        np.testing.assert_array_equal(got, single)


def test_bench_gpus_n_becomes_its_own_launcher():
    """`python bench.py --gpus 4` with no WORLD_SIZE around it re-executes itself under torch.distributed.run with
    one process per GPU on 127.0.0.1 (checked here as the command it would exec; the run itself is
    tests/test_gpu_two_ranks.py::test_bench_gpus_2_launches_its_own_ranks on the GPU)."""
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["PEAQ_BENCH_LAUNCH_DRYRUN"] = "1"
    p = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "4", "--steps", "2"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr[-2000:]
    cmd = json.loads(p.stdout.strip().splitlines()[-1])["launch"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 1024
    assert cmd[-4:] == ["--gpus", "4", "--steps", "2"] and cmd[-5].endswith("bench.py")
