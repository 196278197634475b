#!/usr/bin/env python
"""Per-step kernel timeline from a rocprofv3 --kernel-trace CSV: every launch of ONE steady-state training step
(start offset, duration, idle gap before it) plus per-kernel sums over the last N steps.

    python scripts/step_timeline.py <..._kernel_trace.csv> [n_steps=4]
"""
import csv
import sys


def short(name):
    n = name.split("(")[0]
    n = n.replace("void ", "").replace("pxo::", "")
    if n.startswith("at::native"):
        n = "torch:" + n.split("::")[-1][:40]
    return n[:64]


def main():
    path = sys.argv[1]
    n_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    ends = [i for i, r in enumerate(rows) if "adam" in r[2]]
    if len(ends) < n_steps + 1:
        raise SystemExit("not enough adam launches to delimit steps")
    # a step = (end of the previous step's last pack kernel ...]: delimit at adam launches
    win = rows[ends[-n_steps - 1] + 1: ends[-1] + 1]
    span = win[-1][1] - win[0][0]
    busy = sum(e - s for s, e, _ in win)
    print(f"window: last {n_steps} steps, {span / n_steps / 1e6:.4f} ms per step, {len(win) / n_steps:.1f} launches per step, "
          f"busy {100 * busy / span:.1f} %")
    by = {}
    prev_end = win[0][1]
    for i, (s, e, name) in enumerate(win):
        k = short(name)
        a = by.setdefault(k, [0, 0, 0])
        a[0] += e - s; a[1] += 1
        if i:
            a[2] += max(0, s - prev_end)
        prev_end = max(prev_end, e)
    print(f"{'kernel':64s} {'n/step':>7s} {'us/step':>9s} {'gap us/step':>11s}")
    tot_k = tot_g = 0
    for k, (d, n, g) in sorted(by.items(), key=lambda kv: -kv[1][0]):
        print(f"{k:64s} {n / n_steps:7.2f} {d / n_steps / 1e3:9.1f} {g / n_steps / 1e3:11.1f}")
        tot_k += d; tot_g += g
    print(f"{'TOTAL':64s} {len(win) / n_steps:7.2f} {tot_k / n_steps / 1e3:9.1f} {tot_g / n_steps / 1e3:11.1f}")
    # one step in launch order
    one = rows[ends[-2] + 1: ends[-1] + 1]
    t0 = one[0][0]
    print("\none step in launch order (offset us, duration us, gap-before us):")
    prev_end = None
    for s, e, name in one:
        gap = 0 if prev_end is None else s - prev_end
        print(f"  {(s - t0) / 1e3:9.1f} {(e - s) / 1e3:9.1f} {gap / 1e3:8.1f}  {short(name)}")
        prev_end = e if prev_end is None else max(prev_end, e)


if __name__ == "__main__":
    main()
