#!/usr/bin/env python3
"""Reads a rocprofv3 --kernel-trace CSV and says how much the step kernels of different chains overlap in time.

  python scripts/devtools/trace_overlap.py <kernel_trace.csv> [--match step_kernel] [--out summary.json]

For the kernels whose name contains --match: count, average duration, the queues / streams they ran on, the fraction of
the busy time (union of their intervals) during which >= 2 of them were running, and the average number in flight.
"""
import argparse
import csv
import json


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--match", default="step_kernel")
    ap.add_argument("--out", default=None)
    ap.add_argument("--skip", type=int, default=0, help="ignore this many of the earliest matching kernels (set-up, warm-up)")
    a = ap.parse_args()
    rows = list(csv.DictReader(open(a.csv)))
    name_k = "Kernel_Name" if "Kernel_Name" in rows[0] else "Name"
    ks = [r for r in rows if a.match in r[name_k]]
    ks.sort(key=lambda r: int(r["Start_Timestamp"]))
    ks = ks[a.skip:]
    iv = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in ks]
    ev = sorted([(s, 1) for s, _ in iv] + [(e, -1) for _, e in iv])
    busy = multi = area = 0
    depth, last = 0, ev[0][0]
    for t, d in ev:
        dt = t - last
        if depth >= 1:
            busy += dt
            area += dt * depth
        if depth >= 2:
            multi += dt
        depth += d
        last = t
    dur = [e - s for s, e in iv]
    queues = sorted({r.get("Queue_Id", "?") for r in ks})
    streams = sorted({r.get("Stream_Id", "?") for r in ks})
    grids = sorted({r.get("Grid_Size_X", r.get("Grid_Size", "?")) for r in ks})
    out = {"kernels": len(ks), "match": a.match, "avg_duration_us": sum(dur) / len(dur) / 1e3,
           "span_us": (max(e for _, e in iv) - min(s for s, _ in iv)) / 1e3, "busy_us": busy / 1e3,
           "frac_busy_with_2_or_more_in_flight": multi / busy if busy else 0.0,
           "avg_in_flight_while_busy": area / busy if busy else 0.0,
           "queues": queues, "streams": streams, "grid_sizes": grids,
           "span_us_per_kernel": (max(e for _, e in iv) - min(s for s, _ in iv)) / 1e3 / len(ks)}
    print(json.dumps(out))
    if a.out:
        json.dump(out, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
