#!/usr/bin/env python
"""Idle time between kernels from a rocprofv3 --kernel-trace CSV: how much of the step is launch gaps.

    python scripts/gap_analysis.py gpurun_out/prof/<...>_kernel_trace.csv [first_kernel_substring]
"""
import csv
import sys


def main():
    path = sys.argv[1]
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    # steady state: whole steps only -- from the end of the 5th-last Adam launch to the end of the last one
    ends = [i for i, r in enumerate(rows) if "adam" in r[2]]
    if len(ends) >= 5:
        rows = rows[ends[-5] + 1: ends[-1] + 1]
        print(f"window: the last 4 training steps ({(rows[-1][1] - rows[0][0]) / 4e6:.3f} ms per step incl. the feeder)")
    else:
        rows = rows[len(rows) // 3:]
    busy = sum(e - s for s, e, _ in rows)
    span = rows[-1][1] - rows[0][0]
    gaps = []
    prev_end = rows[0][1]
    for s, e, name in rows[1:]:
        gaps.append((max(0, s - prev_end), name))
        prev_end = max(prev_end, e)
    idle = sum(g for g, _ in gaps)
    by = {}
    for g, name in gaps:
        k = name.split("(")[0][-60:]
        a = by.setdefault(k, [0, 0])
        a[0] += g; a[1] += 1
    print(f"kernels {len(rows)}  span {span / 1e6:.2f} ms  busy {busy / 1e6:.2f} ms ({100 * busy / span:.1f} %)  idle {idle / 1e6:.2f} ms")
    print("idle before kernel (top 12):")
    for k, (g, n) in sorted(by.items(), key=lambda kv: -kv[1][0])[:12]:
        print(f"  {g / 1e6:8.3f} ms  {n:5d} x  avg {g / n / 1e3:7.1f} us  {k}")


if __name__ == "__main__":
    main()
