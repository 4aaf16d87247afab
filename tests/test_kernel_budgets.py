"""Register budgets the design depends on, read from the compiler's own kernel metadata (hipcc
cross-compiles for gfx950 without a GPU).  DESIGN.md 3: a SIMD's 512 registers per lane must hold
  * three waves of the front end or of the pattern back end in any mix (<= 170 each, no scratch in the
    front end), and
  * two waves of the filter-bank kernel of the default engine PLUS one of the high-pass kernel, whose walk
    over the next launch's samples runs beside the bank (with 212 + 102 registers the advanced pass was 55 ms
    longer: the bank waited for the high-pass filter)."""
import re
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "gstpeaq_amd" / "csrc"


def kernel_metadata(source, tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not Path(hipcc).exists():
        pytest.skip("no hipcc")
    out = tmp_path / (source + ".s")
    subprocess.run([hipcc, "-O3", "--offload-arch=gfx950", "-std=c++17", f"-I{CSRC}", f"-I{ROOT / 'include'}", "-S",
                    "--cuda-device-only", "-o", str(out), str(CSRC / source)], check=True, capture_output=True)
    meta = {}
    name = None
    for line in out.read_text().splitlines():
        m = re.match(r"\s+\.name:\s+(\S+)", line)
        if m:
            name = m.group(1)
            meta[name] = {}
        m = re.match(r"\s+\.(vgpr_count|vgpr_spill_count|private_segment_fixed_size|group_segment_fixed_size):\s+(\d+)", line)
        if m and name:
            meta[name][m.group(1)] = int(m.group(2))
    return meta


def find(meta, fragment):
    (k, v), = [(k, v) for k, v in meta.items() if fragment in k]
    return v


def test_filter_bank_leaves_room_for_the_high_pass_walk(tmp_path):
    meta = kernel_metadata("peaq_fb.hip", tmp_path)
    bank, hp = find(meta, "fb_bank_kernel_h3"), find(meta, "fb_hp_kernel")
    assert bank["vgpr_spill_count"] == 0 and hp["vgpr_spill_count"] == 0
    assert 2 * bank["vgpr_count"] + hp["vgpr_count"] <= 512, (bank, hp)
    assert 2 * bank["group_segment_fixed_size"] <= 160 * 1024          # two workgroups per CU


def test_three_waves_per_simd_for_the_fft_path(tmp_path):
    fe = kernel_metadata("peaq_frontend.hip", tmp_path)
    be = kernel_metadata("peaq_backend.hip", tmp_path)
    for k in ("frontend_kernelILi109E", "frontend_kernelILi55E"):
        v = find(fe, k)
        assert v["vgpr_count"] <= 170 and v["vgpr_spill_count"] == 0 and v["private_segment_fixed_size"] == 0, (k, v)
    v = find(be, "backend_kernelILi109ELb0ELb0E")
    assert v["vgpr_count"] <= 170 and v["vgpr_spill_count"] == 0, v
