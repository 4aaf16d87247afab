#!/usr/bin/env python3
"""Live-pipeline broker throughput (BASELINE.json configs[5]): N concurrent sessions, each fed a
`seconds` long stereo stream in GStreamer-sized buffers (4096 samples) by feeder threads while the
broker's tick thread batches whatever became ready.  Host buffers in, so this rate includes the
PCIe copies and the host FIFO work -- it is NOT bench.py's `value`.
Prints one JSON line."""
import argparse
import json
import sys
import threading
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sessions", type=int, default=1024)
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--buffer", type=int, default=4096)
    ap.add_argument("--feeders", type=int, default=8)
    ap.add_argument("--period-us", type=int, default=2000)
    ap.add_argument("--no-broker", action="store_true", help="one peaq_session per stream instead")
    ap.add_argument("--advanced", action="store_true")
    args = ap.parse_args()
    import gstpeaq_amd
    import synth_np
    ctx = gstpeaq_amd.Context(0)
    n = int(args.seconds * 48000)
    # 16 distinct seeded pairs, reused round-robin (the host generator is slow)
    pairs = [synth_np.pair(1 + i, 2, n) for i in range(16)]
    if args.no_broker:
        sessions = [gstpeaq_amd.Session(ctx, args.advanced, 2) for _ in range(args.sessions)]
    else:
        b = gstpeaq_amd.Broker(ctx, 2, args.sessions, advanced=args.advanced)
        sids = [b.open() for _ in range(args.sessions)]
        b.start(args.period_us)
    results = [None] * args.sessions

    def feeder(w):
        mine = range(w, args.sessions, args.feeders)
        for pos in range(0, n, args.buffer):
            for i in mine:
                ref, test = pairs[i % 16]
                if args.no_broker:
                    sessions[i].push(0, ref[pos:pos + args.buffer])
                    sessions[i].push(1, test[pos:pos + args.buffer])
                else:
                    b.push(sids[i], 0, ref[pos:pos + args.buffer])
                    b.push(sids[i], 1, test[pos:pos + args.buffer])
        for i in mine:
            if args.no_broker:
                sessions[i].flush()
                results[i] = sessions[i].results()
            else:
                b.flush(sids[i])
        if not args.no_broker:
            for i in mine:
                results[i] = b.results(sids[i])

    t0 = time.perf_counter()
    th = [threading.Thread(target=feeder, args=(w,)) for w in range(args.feeders)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    frames = sum(r["frames"] for r in results)
    line = dict(sessions=args.sessions, seconds=args.seconds, buffer=args.buffer, wall_s=dt,
                frame_pairs_per_s=frames / dt, realtime_streams=args.sessions * args.seconds / dt,
                mode=("one session per stream" if args.no_broker else "broker") + (", advanced" if args.advanced else ", basic"),
                odg_mean=float(np.mean([r["odg"] for r in results])))
    if not args.no_broker:
        line.update(b.stats())
        b.stop()
    print(json.dumps(line))


if __name__ == "__main__":
    main()
