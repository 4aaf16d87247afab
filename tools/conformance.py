#!/usr/bin/env python3
"""Conformance runner: the job of the reference's src/checkconformanceresults.sh
and doc/make_conformance_tables.sh on the MI355X engine.

Given the directory of the 16 ITU-R BS.1387 conformance items (<x>cod<yyy>.wav with its
<x>ref<yyy>.wav, 48 kHz), computes DI/ODG of every item in the basic and the advanced version --
all 16 pairs of a version in ONE batched launch -- and prints, per item, the DI next to the DI the
reference implementation reaches ("Actual DI" of doc/conformance_*_table.xml; equality at three
decimals is the pass criterion of checkconformanceresults.sh:24-31) and the ITU values, then the
bias / mean square error against ITU (make_conformance_tables.sh:78-81).

  CONFORMANCEDATADIR=/path/to/items python tools/conformance.py [--cli] [--mode basic|advanced|both] [--gpus N]

--gpus N deals a version's items out to N GPUs (a contiguous block of items per device, gstpeaq_amd.parallel.shard:
pairs are closed computations, nothing is exchanged) and runs the devices' launches from N threads; --devices
0,0 names the ordinals explicitly (an ordinal may repeat).

Exit status like the reference script: 77 when the data are absent (test NOT run), 1 on a
mismatch, 0 when every item matches.  --cli runs the `peaq` command-line tool per item instead of
the batch API (the reference's way)."""
import argparse
import json
import os
import re
import subprocess
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
TABLES = ROOT / "tests" / "golden" / "conformance_tables.json"


def item_files(datadir, item):
    cod = Path(datadir) / f"{item}.wav"
    ref = Path(datadir) / f"{item.replace('cod', 'ref', 1)}.wav"
    return ref, cod


def run_batch(ctx, pairs, advanced):
    """pairs: list of (ref, test) float32 [n, ch]; one batched launch on ctx's device -> list of result dicts"""
    import torch
    import gstpeaq_amd
    if not pairs:
        return []
    dev = torch.device("cuda", int(ctx.device))
    ch = pairs[0][0].shape[1]
    stride = max(max(len(r), len(t)) for r, t in pairs)
    stride += stride & 1
    ref = np.zeros((len(pairs), stride, ch), np.float32)
    test = np.zeros_like(ref)
    n_ref = np.zeros(len(pairs), np.uint32)
    n_test = np.zeros(len(pairs), np.uint32)
    for i, (r, t) in enumerate(pairs):
        ref[i, :len(r)], test[i, :len(t)] = r, t
        n_ref[i], n_test[i] = len(r), len(t)
    return gstpeaq_amd.batch_run(ctx, advanced, torch.from_numpy(ref).to(dev), torch.from_numpy(test).to(dev),
                                 n_ref, n_test)


def run_sharded(ctxs, pairs, advanced):
    """the pairs dealt out to the contexts' devices (contiguous blocks), one launch per device, side by side"""
    from concurrent.futures import ThreadPoolExecutor
    from gstpeaq_amd import parallel
    if len(ctxs) == 1:
        return run_batch(ctxs[0], pairs, advanced)
    blocks = [parallel.shard(len(pairs), r, len(ctxs)) for r in range(len(ctxs))]
    with ThreadPoolExecutor(len(ctxs)) as pool:
        parts = list(pool.map(lambda a: run_batch(a[0], pairs[a[1][0]:a[1][1]], advanced), zip(ctxs, blocks)))
    return [res for part in parts for res in part]


def run_cli(ref, cod, advanced):
    cli = ROOT / "gstpeaq_amd" / "cli" / "peaq"
    out = subprocess.run([str(cli), "--advanced" if advanced else "--basic", str(ref), str(cod)],
                         capture_output=True, text=True, env=dict(os.environ, LC_ALL="C"))
    if out.returncode != 0:
        raise RuntimeError(out.stderr.strip())
    odg = float(re.search(r"Objective Difference Grade: (\S+)", out.stdout).group(1))
    di = float(re.search(r"Distortion Index: (\S+)", out.stdout).group(1))
    return dict(di=di, odg=odg)


def check(mode, items, datadir, use_cli, ctxs):
    advanced = mode == "advanced"
    print(f"{mode.capitalize()} version:")
    if use_cli:
        results = [run_cli(*item_files(datadir, it["item"]), advanced) for it in items]
    else:
        from gstpeaq_amd.wavio import read_wav
        by_ch = {}
        for idx, it in enumerate(items):
            (r, fr), (t, ft) = (read_wav(p) for p in item_files(datadir, it["item"]))
            if fr != 48000 or ft != 48000:
                raise SystemExit(f"{it['item']}: 48 kHz input required (got {fr}/{ft})")
            if r.shape[1] != t.shape[1] or r.shape[1] > 2:
                raise SystemExit(f"{it['item']}: mono or stereo, same layout on both files")
            by_ch.setdefault(r.shape[1], []).append((idx, r, t))
        results = [None] * len(items)
        for ch, group in by_ch.items():                      # one launch per channel layout
            for (idx, _, _), res in zip(group, run_sharded(ctxs, [(r, t) for _, r, t in group], advanced)):
                results[idx] = res
    ok = True
    d_odg, d_di = [], []
    for it, res in zip(items, results):
        di3 = f"{res['di']:.3f}"
        same = di3 == it["reference_di"]
        ok &= same
        d_di.append(res["di"] - float(it["itu_di"]))
        d_odg.append(res["odg"] - float(it["itu_odg"]))
        print(f"{it['item']} DI {di3} (reference implementation {it['reference_di']}, ITU {it['itu_di']})  "
              f"ODG {res['odg']:.3f} (ITU {it['itu_odg']})  {'OK' if same else 'FAILED'}")
    print(f"ODG mean error (bias): {np.mean(d_odg):.3f}")
    print(f"ODG mean square error: {np.mean(np.square(d_odg)):.6f}")
    print(f"DI mean error (bias): {np.mean(d_di):.3f}")
    print(f"DI mean square error: {np.mean(np.square(d_di)):.6f}")
    return ok


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--mode", choices=["basic", "advanced", "both"], default="both")
    ap.add_argument("--cli", action="store_true")
    ap.add_argument("--datadir", default=os.environ.get("CONFORMANCEDATADIR"))
    ap.add_argument("--gpus", type=int, default=1, help="deal the items out to this many GPUs (ordinals 0..N-1)")
    ap.add_argument("--devices", default=None, help="explicit device ordinals, e.g. 0,1,2,3 (overrides --gpus)")
    args = ap.parse_args()
    if not args.datadir:
        print("CONFORMANCEDATADIR not set, conformance test NOT run.")
        return 77
    if not Path(args.datadir).is_dir():
        print("Reference data not found, conformance test NOT run.")
        return 77
    tables = json.loads(TABLES.read_text())
    modes = ["basic", "advanced"] if args.mode == "both" else [args.mode]
    missing = [str(p) for m in modes for it in tables[m] for p in item_files(args.datadir, it["item"]) if not p.exists()]
    if missing:
        print(f"Reference data incomplete ({len(set(missing))} files missing, e.g. {missing[0]}), "
              "conformance test NOT run.")
        return 77
    ctxs = None
    if not args.cli:
        import gstpeaq_amd
        devices = [int(d) for d in args.devices.split(",")] if args.devices else list(range(max(args.gpus, 1)))
        ctxs = [gstpeaq_amd.Context(d) for d in devices]
    ok = True
    for m in modes:
        ok &= check(m, tables[m], args.datadir, args.cli, ctxs)
    return 0 if ok else 1


if __name__ == "__main__":
    try:
        sys.exit(main())
    except Exception as e:                                   # noqa: BLE001 -- 1 is reserved for "an item differs"
        print(f"conformance run failed: {e}", file=sys.stderr)
        sys.exit(2)
