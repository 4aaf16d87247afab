#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd databases (…_results.db) into the small text/JSON
files committed under profiles/.

  tools/rocprof_summary.py stats  DB              -> per-kernel calls / total / average (like --stats)
  tools/rocprof_summary.py pmc    DB [DB ...]     -> per-kernel average of every collected counter
  tools/rocprof_summary.py timeline DB [N]        -> the last N kernel dispatches in start order: name, queue, start and
                                                     end in ms after the first of them
"""
import json
import sqlite3
import sys


def short(name):
    name = name.replace("void ", "")
    return name if len(name) < 90 else name[:87] + "..."


def stats(db):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    out = [dict(kernel=short(n), calls=c, total_us=t / 1e3 if t > 1e6 else t, avg_us=a, pct=p) for n, c, t, a, p in rows]
    # durations in top_kernels are microseconds
    out = [dict(kernel=short(n), calls=c, total_us=round(t, 1), avg_us=round(a, 2), pct=round(p, 3))
           for n, c, t, a, p in rows]
    extra = cur.execute("select name, vgpr_count, accum_vgpr_count, sgpr_count, lds_size, scratch_size, "
                        "workgroup_x, min(grid_x), max(grid_x) from kernels group by name").fetchall()
    res = {short(n): dict(vgpr=v, agpr=a, sgpr=s, lds=l, scratch=sc, workgroup=w, grid_min=g0, grid_max=g1)
           for n, v, a, s, l, sc, w, g0, g1 in extra}
    return dict(kernels=out, resources=res)


def pmc(dbs):
    out = {}
    for db in dbs:
        cur = sqlite3.connect(db).cursor()
        rows = cur.execute("select kernel_name, counter_name, count(*), avg(value), sum(value), avg(duration) "
                           "from counters_collection group by kernel_name, counter_name").fetchall()
        for k, c, n, avg, tot, dur in rows:
            out.setdefault(short(k), {})[c] = dict(dispatches=n, avg=avg, sum=tot, avg_duration_ns=dur)
    return out


def timeline(db, n):
    cur = sqlite3.connect(db).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)").fetchall()]
    q = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
    rows = cur.execute(f"select name, {q}, start, end from kernels order by start").fetchall()[-n:]
    t0 = rows[0][2]
    return [dict(kernel=short(k)[:60], queue=qq, start_ms=round((a - t0) / 1e6, 3), end_ms=round((b - t0) / 1e6, 3),
                 ms=round((b - a) / 1e6, 3)) for k, qq, a, b in rows]


if __name__ == "__main__":
    mode = sys.argv[1]
    if mode == "timeline":
        for r in timeline(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 40):
            print("%-62s q%-3s %9.3f .. %9.3f  (%8.3f ms)" % (r["kernel"], r["queue"], r["start_ms"], r["end_ms"], r["ms"]))
    elif mode == "stats":
        print(json.dumps(stats(sys.argv[2]), indent=1))
    else:
        print(json.dumps(pmc(sys.argv[2:]), indent=1))
