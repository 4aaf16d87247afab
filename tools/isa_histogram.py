#!/usr/bin/env python3
"""Static opcode-class histogram of one kernel of a hipcc -S listing (development tool).
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -Igstpeaq_amd/csrc -Iinclude -S --cuda-device-only -o fe.s gstpeaq_amd/csrc/peaq_frontend.hip
  tools/isa_histogram.py fe.s frontend_kernelILi109E
Classes: FP64 arithmetic, DPP moves, selects, plain moves, lane reads, permlane swaps, compares, other integer /
FP32 vector instructions, LDS, vector memory, scalar.  Static counts: both waves' roles and both settings of a
run-time switch are in the listing, a wave executes about 60 % of it (the PMC profile has the dynamic count)."""
import collections
import re
import sys


def classify(t):
    op = t.split()[0]
    if op.startswith("v_"):
        if "dpp" in t:
            return "valu dpp move" if op.startswith("v_mov") else "valu dpp arithmetic"
        if op.startswith("v_mfma"):
            return "mfma"
        if "f64" in op:
            return "valu fp64"
        if op.startswith("v_cndmask"):
            return "valu select"
        if op.startswith("v_mov") or op.startswith("v_accvgpr"):
            return "valu move"
        if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
            return "valu lane read/write"
        if op.startswith("v_permlane"):
            return "valu permlane swap"
        if op.startswith("v_cmp"):
            return "valu compare"
        return "valu other (integer, address, fp32, conversions)"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("buffer_", "global_", "scratch_", "flat_")):
        return "vector memory"
    if op.startswith("s_"):
        return "scalar"
    return "other"


def main():
    text = open(sys.argv[1]).read().split("\n")
    frag = sys.argv[2]
    start = next(i for i, l in enumerate(text) if re.match(r"^_Z\S*" + re.escape(frag) + r"\S*:", l))
    end = next(i for i in range(start, len(text)) if text[i].startswith(".Lfunc_end"))
    cls, ops = collections.Counter(), collections.Counter()
    for l in text[start + 1:end + 1]:
        t = l.strip()
        if not t or t[0] in ";." or t.endswith(":"):
            continue
        k = classify(t)
        cls[k] += 1
        if k.startswith("valu other") or k in ("valu select", "valu move"):
            ops[t.split()[0]] += 1
    total_v = sum(v for k, v in cls.items() if k.startswith("valu") or k == "mfma")
    print(f"kernel {frag}: {sum(cls.values())} instructions, {total_v} vector ALU")
    for k, v in cls.most_common():
        print(f"  {v:6d}  {k}" + (f"   ({100 * v / total_v:.1f} % of the vector ALU instructions)" if k.startswith('valu') else ""))
    print("  most frequent non-FP64 vector opcodes:", ", ".join(f"{k} {v}" for k, v in ops.most_common(14)))


if __name__ == "__main__":
    main()
