#!/usr/bin/env python
"""Condense one GPU session's rocprofv3 output (gpurun_out/) into profiles/<tag>_*.

  kernel stats  : rocprofv3 --kernel-trace --stats  -> <tag>_kernel_stats.csv (top rows)
  PMC passes    : rocprofv3 --kernel-trace --pmc ... (separate passes) -> <tag>_pmc.md
Corrections (MI355X_MICROARCH.md): SQ_* / GRBM_* counters arrive summed over the 8 XCDs;
SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles; FETCH_SIZE under-reports wide
coalesced reads by 2x on gfx950 (calibrated here on wgrad_kernel<256,256>: 2 x 1 KiB/row read).
"""
import collections
import csv
import os
import sys


def load_pmc(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    seen = set()
    dur = collections.defaultdict(float)
    with open(path) as f:
        for r in csv.DictReader(f):
            k = r["Kernel_Name"]
            if "pxo::" not in k:
                continue
            k = k.split("(")[0].replace("void ", "")
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            key = (k, r["Dispatch_Id"])
            if key not in seen:
                seen.add(key)
                cnt[k] += 1
                dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    return agg, cnt, dur


def main(src, dst_dir, tag):
    os.makedirs(dst_dir, exist_ok=True)
    ks = os.path.join(src, "prof", "bench_kernel_stats.csv")
    if os.path.exists(ks):
        with open(ks) as f, open(os.path.join(dst_dir, f"{tag}_kernel_stats.csv"), "w") as g:
            for i, line in enumerate(f):
                if i <= 24:
                    g.write(line if len(line) < 400 else line[:200] + '...",' + ",".join(line.rsplit(",", 7)[1:]))
    rows = {}
    for i in range(1, 9):
        p = os.path.join(src, f"pmc{i}", "pmc_counter_collection.csv")
        if not os.path.exists(p):
            continue
        agg, cnt, dur = load_pmc(p)
        for k in agg:
            rows.setdefault(k, {})
            for c, v in agg[k].items():
                rows[k][c] = v / cnt[k]
            rows[k]["_sec"] = dur[k] / cnt[k]
            rows[k]["_n"] = cnt[k]
    out = [f"# PMC summary `{tag}` (per launch averages over coarse+fine launches; python bench.py --steps 2 --warmup 1)\n",
           "| kernel | launches | avg ms | clock GHz | MFMA busy | waitcnt/barrier | issue-stall | active issue | VALU/MFMA insts | LDS conflict | HBM read MB (FETCH x2) | HBM write MB |",
           "|---|---|---|---|---|---|---|---|---|---|---|---|"]
    for k, r in sorted(rows.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", 0)):
        if "SQ_WAVE_CYCLES" not in r:
            continue
        gui = r.get("GRBM_GUI_ACTIVE", 0) / 8.0            # per-XCD cycles
        sec = r["_sec"]
        clock = gui / sec / 1e9 if sec else 0
        simd_cycles = gui * 1024 if gui else float("nan")   # 256 CUs x 4 SIMDs
        wave = r["SQ_WAVE_CYCLES"] * 4
        f = lambda x: f"{100 * x:.1f}%"
        out.append("| `{}` | {} | {:.3f} | {:.2f} | {} | {} | {} | {} | {:.2f} | {} | {:.0f} | {:.0f} |".format(
            k[:70], int(r["_n"]), sec * 1e3, clock, f(r["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles),
            f(r["SQ_WAIT_ANY"] * 4 / wave), f(r["SQ_WAIT_INST_ANY"] * 4 / wave), f(r["SQ_ACTIVE_INST_ANY"] * 4 / wave),
            r["SQ_INSTS_VALU"] / max(r["SQ_INSTS_MFMA"], 1),
            f(r.get("SQ_LDS_BANK_CONFLICT", 0) / max(r.get("SQ_LDS_IDX_ACTIVE", 1), 1)),
            r.get("FETCH_SIZE", 0) * 2 / 1024, r.get("WRITE_SIZE", 0) / 1024))
    out.append("\nMFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 x 1024 SIMDs); the wave-cycle shares are "
               "SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES (disjoint).  FETCH_SIZE/WRITE_SIZE are KiB.")
    with open(os.path.join(dst_dir, f"{tag}_pmc.md"), "w") as g:
        g.write("\n".join(out) + "\n")
    print("\n".join(out))
    # HBM bytes per launch for bench.py's roofline.traffic (FETCH_SIZE x2: the gfx950 under-count of wide coalesced
    # reads, MI355X_MICROARCH.md "HBM"; both counters are KiB)
    traffic = {"source": f"profiles/{tag}_pmc.md: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over `python "
                         "bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras`; FETCH_SIZE x2 (gfx950 under-count of "
                         "wide coalesced reads, MI355X_MICROARCH.md), KiB units, averaged over the coarse and fine launches",
               "kernels": {}}
    for k, r in rows.items():
        if "FETCH_SIZE" in r and "WRITE_SIZE" in r:
            name = k.replace("pxo::", "").split("<")[0]
            ent = traffic["kernels"].setdefault(name, {"hbm_read_bytes_per_launch": 0, "hbm_write_bytes_per_launch": 0,
                                                       "_launches": 0})
            n = r["_n"]          # several template instances of one kernel name: launch-weighted average
            ent["hbm_read_bytes_per_launch"] += r["FETCH_SIZE"] * 2 * 1024 * n
            ent["hbm_write_bytes_per_launch"] += r["WRITE_SIZE"] * 1024 * n
            ent["_launches"] += n
    for ent in traffic["kernels"].values():
        n = ent.pop("_launches")
        ent["hbm_read_bytes_per_launch"] = int(ent["hbm_read_bytes_per_launch"] / n)
        ent["hbm_write_bytes_per_launch"] = int(ent["hbm_write_bytes_per_launch"] / n)
    if traffic["kernels"]:
        import json
        import subprocess
        try:                     # which code the counters were collected on (the session ran the working tree of this commit)
            traffic["commit"] = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], stderr=subprocess.DEVNULL,
                                                        cwd=os.path.dirname(os.path.abspath(__file__))).decode().strip()
        except Exception:
            pass
        with open(os.path.join(dst_dir, "hbm_traffic.json"), "w") as g:
            json.dump(traffic, g, indent=1)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out", sys.argv[2] if len(sys.argv) > 2 else "profiles",
         sys.argv[3] if len(sys.argv) > 3 else "r01")
