#!/usr/bin/env python3
"""profiles/pmc_frontend.json (read by bench.py for roofline.traffic) from the two rocprofv3 --pmc
passes summarised by tools/collect_profiles.sh (gpurun_out/pmc_hbm_basic.json)."""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
src = Path(sys.argv[1] if len(sys.argv) > 1 else ROOT / "gpurun_out" / "pmc_hbm_basic.json")
d = json.loads(src.read_text())
fe = next(v for k, v in d.items() if "frontend_kernel<109>" in k)
be = next(v for k, v in d.items() if "backend_kernel<109" in k)
launches = fe["FETCH_SIZE"]["dispatches"]
pairs, frames, algo = 4096, 468, 16384
algo_launch = pairs * frames * algo / launches
rd = fe["FETCH_SIZE"]["avg"] * 1024 * 2          # gfx950: FETCH_SIZE reports 1/2 of a wide coalesced stream
wr = fe["WRITE_SIZE"]["avg"] * 1024
out = {
    "_command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python bench.py "
                "--steps 1 --warmup 0 --no-cpu-baseline (configs[1])",
    "_units": "bytes per dispatch, averaged over the front-end launches of one pass; FETCH_SIZE (KiB) doubled "
              "(gfx950 correction, MI355X_MICROARCH.md HBM section), WRITE_SIZE (KiB) as is",
    "kernel": "frontend_kernel<109>",
    "launches": launches,
    "algorithmic_bytes_per_launch": algo_launch,
    "hbm_read_bytes_per_launch": rd,
    "hbm_write_bytes_per_launch": wr,
    "hbm_bytes_per_launch": rd + wr,
    "read_over_algorithmic": rd / algo_launch,
    "record_bytes_per_launch": pairs * frames * 2 * 4608 / launches,
    "note": "reads = the input samples once (50 % frame overlap and channel interleave absorbed by the per-XCD L2); "
            "writes = the per-frame records handed to the back end (4608 B per frame and channel); no scratch traffic",
    "backend_kernel<109,false>": {
        "hbm_read_bytes_per_launch": be["FETCH_SIZE"]["avg"] * 1024 * 2,
        "hbm_write_bytes_per_launch": be["WRITE_SIZE"]["avg"] * 1024,
    },
    "counters": {k: v for k, v in d.items() if "peaq::" in k},
}
(ROOT / "profiles" / "pmc_frontend.json").write_text(json.dumps(out, indent=1) + "\n")
print(json.dumps({k: v for k, v in out.items() if k != "counters"}, indent=1))
