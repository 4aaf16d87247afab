#!/usr/bin/env python3
"""profiles/pmc_frontend.json -- the counter-derived figures bench.py quotes next to its live timing
(roofline.traffic, roofline.valu_issue_frac), with the commit and kernel time they were taken at.
Inputs: the rocprofv3 --pmc summaries written on the GPU box by tools/collect_profiles.sh
(gpurun_out/pmc_hbm_basic.json: FETCH_SIZE / WRITE_SIZE, configs[1]) and tools/pmc_mix.sh
(gpurun_out/pmc_mix.json: instruction mix, 1024 pairs) and gpurun_out/stats_basic.json."""
import json
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
G = ROOT / "gpurun_out"
d = json.loads((G / "pmc_hbm_basic.json").read_text())
mix = json.loads((G / "pmc_mix.json").read_text())
stats = json.loads((G / "stats_basic.json").read_text())
fe = next(v for k, v in d.items() if "frontend_kernel<109>" in k)
be = next(v for k, v in d.items() if "backend_kernel<109" in k)
fm = next(v for k, v in mix.items() if "frontend_kernel<109>" in k)
bm = next(v for k, v in mix.items() if "backend_kernel<109" in k)
launches = fe["FETCH_SIZE"]["dispatches"]
pairs, frames, algo = 4096, 468, 16384
algo_launch = pairs * frames * algo / launches
rd = fe["FETCH_SIZE"]["avg"] * 1024 * 2          # gfx950: FETCH_SIZE reports 1/2 of a wide coalesced stream
wr = fe["WRITE_SIZE"]["avg"] * 1024
w = fm["SQ_WAVES"]["avg"]
fp64 = sum(fm[c]["avg"] for c in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64",
                                  "SQ_INSTS_VALU_TRANS_F64")) / w
sys.path.insert(0, str(ROOT))
from bench import source_hash  # noqa: E402
# back end: instructions per wave and FRAME (a wave walks all 468 frames of its pair and channel over the launches)
be_wave_frames = bm["SQ_WAVES"]["avg"] * frames
be_valu = bm["SQ_INSTS_VALU"]["sum"] / be_wave_frames
be_fp64 = sum(bm[c]["sum"] for c in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64",
                                     "SQ_INSTS_VALU_TRANS_F64")) / be_wave_frames
commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
kavg = next(k["avg_us"] for k in stats["kernels"] if "frontend_kernel<109>" in k["kernel"]) / 1e3
out = {
    "_command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python bench.py "
                "--steps 1 --warmup 0 --no-cpu-baseline --no-advanced (configs[1]); instruction mix: tools/pmc_mix.sh (1024 pairs)",
    "_units": "bytes per dispatch, averaged over the front-end launches of one pass; FETCH_SIZE (KiB) doubled "
              "(gfx950 correction, MI355X_MICROARCH.md HBM section), WRITE_SIZE (KiB) as is",
    "commit": commit,
    "source_hash": source_hash(),
    "kernel": "frontend_kernel<109>",
    "kernel_avg_ms": kavg,
    "launches": launches,
    "algorithmic_bytes_per_launch": algo_launch,
    "hbm_read_bytes_per_launch": rd,
    "hbm_write_bytes_per_launch": wr,
    "hbm_bytes_per_launch": rd + wr,
    "read_over_algorithmic": rd / algo_launch,
    "traffic_over_algorithmic": (rd + wr) / algo_launch,
    "record_bytes_per_launch": pairs * frames * 2 * 2816 / launches,
    "valu_insts_per_wave": fm["SQ_INSTS_VALU"]["avg"] / w,
    "valu_fp64_insts_per_wave": fp64,
    "lds_insts_per_wave": fm["SQ_INSTS_LDS"]["avg"] / w,
    "vmem_rd_insts_per_wave": fm["SQ_INSTS_VMEM_RD"]["avg"] / w,
    "wave_cycles_per_wave": fm["SQ_WAVE_CYCLES"]["avg"] / w * 4,
    "wait_any_frac": fm["SQ_WAIT_ANY"]["avg"] / fm["SQ_WAVE_CYCLES"]["avg"],
    "note": "reads = the input samples once (50 % frame overlap and channel interleave absorbed by the per-XCD L2, "
            "plus the L2 prefetch touching every line once more from L2); writes = the per-frame records handed to the "
            "back end (2816 B per frame and channel); no scratch traffic",
    "backend_kernel<109,false>": {
        "hbm_read_bytes_per_launch": be["FETCH_SIZE"]["avg"] * 1024 * 2,
        "hbm_write_bytes_per_launch": be["WRITE_SIZE"]["avg"] * 1024,
        "valu_insts_per_wave_frame": be_valu,
        "valu_fp64_insts_per_wave_frame": be_fp64,
        "lds_insts_per_wave_frame": bm["SQ_INSTS_LDS"]["sum"] / be_wave_frames,
    },
}
# the FP64 filter bank (tools/pmc_fb.sh -> gpurun_out/pmc_fb.json, 1024 pairs x 10 s = 2500 blocks per signal):
# instructions per wave and filter-bank block, what bench.py's advanced.roofline.simd_busy_frac is computed from
fbp = G / "pmc_fb.json"
if fbp.exists():
    from bench import fb_source_hash  # noqa: E402
    fb = json.loads(fbp.read_text())
    bk = next((v for k, v in fb.items() if "fb_bank_kernel<peaq::MfmaF64>" in k), None)
    if bk:
        blocks = 2500.0
        w4 = 1024 * 2 * 2 * 4                          # waves of one pass: signals x 4 (every launch starts them anew)
        tot = lambda c: bk[c]["sum"] / w4 / blocks if c in bk else 0.0   # noqa: E731
        f64 = sum(tot(c) for c in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64"))
        out["fb_bank_kernel<MfmaF64>"] = {
            "commit": commit, "source_hash": fb_source_hash(),
            "valu_per_wave_block": tot("SQ_INSTS_VALU") - tot("SQ_INSTS_MFMA"),
            "valu_fp64_per_wave_block": f64,
            "mfma_per_wave_block": tot("SQ_INSTS_MFMA"),
            "lds_per_wave_block": tot("SQ_INSTS_LDS"),
            "lds_atomic_per_wave_block": tot("SQ_INSTS_LDS_ATOMIC"),
            "lds_idx_active_cycles_per_wave_block": tot("SQ_LDS_IDX_ACTIVE"),
            "lds_bank_conflict_cycles_per_wave_block": tot("SQ_LDS_BANK_CONFLICT"),
            "wave_cycles_per_wave_block": tot("SQ_WAVE_CYCLES") * 4,
            "note": "a tile is 10 blocks; SQ_INSTS_VALU counts the matrix instructions too (taken out above)",
        }
(ROOT / "profiles" / "pmc_frontend.json").write_text(json.dumps(out, indent=1) + "\n")
print(json.dumps(out, indent=1))
