#!/usr/bin/env python
"""Condense the rocprofv3 output of scripts/octree_bench.py (gpurun_out/oprof, opmc1, opmc2) into
profiles/<tag>_octree_kernels.md: per-kernel launches, average duration, HBM bytes per launch (PMC
FETCH_SIZE x2 per the gfx950 correction in MI355X_MICROARCH.md, WRITE_SIZE) and the bandwidth they imply."""
import collections
import csv
import os
import sys


def pmc(path, counter):
    tot, n, dur, seen = collections.defaultdict(float), collections.Counter(), collections.defaultdict(float), set()
    if not os.path.exists(path):
        return {}, {}
    with open(path) as f:
        for r in csv.DictReader(f):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if "pxo::" not in k or r["Counter_Name"] != counter:
                continue
            tot[k] += float(r["Counter_Value"])
            key = (k, r["Dispatch_Id"])
            if key not in seen:
                seen.add(key); n[k] += 1
                dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
    return {k: tot[k] / n[k] for k in tot}, {k: dur[k] / n[k] for k in dur}


def main(src, dst_dir, tag, pmc_cams=2):
    import json
    # the scene the counter passes ran on (scripts/octree_bench.py prints it; gpu_session.sh keeps each pass's JSON): a ratio of
    # these bytes to a floor is only meaningful for the SAME tree, cameras and image size -- octree_bench.py checks before dividing
    scenes = []
    for name in ("opmc1.json", "opmc2.json"):
        try:
            with open(os.path.join(src, name)) as f:
                scenes.append(json.load(f)["scene"])
        except Exception:
            pass
    if len(scenes) == 2 and scenes[0] != scenes[1]:
        raise SystemExit(f"the two counter passes ran on different scenes: {scenes}")
    stats = {}
    ks = os.path.join(src, "oprof", "obench_kernel_stats.csv")
    with open(ks) as f:
        for r in csv.DictReader(f):
            k = r["Name"].split("(")[0].replace("void ", "")
            if "pxo::" in k:
                stats[k] = (int(r["Calls"]), float(r["AverageNs"]) * 1e-6, float(r["Percentage"]))
    rd, _ = pmc(os.path.join(src, "opmc1", "pmc_counter_collection.csv"), "FETCH_SIZE")
    wr, _ = pmc(os.path.join(src, "opmc2", "pmc_counter_collection.csv"), "WRITE_SIZE")
    out = [f"# PlenOctree-side kernels `{tag}` (rocprofv3 --kernel-trace --stats of scripts/octree_bench.py; HBM bytes from",
           "# separate --pmc FETCH_SIZE / WRITE_SIZE passes, KiB, FETCH x2 on gfx950)\n",
           "| kernel | launches | avg ms | % of GPU time | HBM read MB / launch | HBM write MB / launch | HBM GB/s |", "|---|---|---|---|---|---|---|"]
    for k, (calls, ms, pct) in sorted(stats.items(), key=lambda kv: -kv[1][2]):
        r = rd.get(k, float("nan")) * 2 / 1024
        w = wr.get(k, float("nan")) / 1024
        out.append(f"| `{k[:60]}` | {calls} | {ms:.3f} | {pct:.1f} | {r:.1f} | {w:.1f} | {(r + w) / ms:.0f} |")
    os.makedirs(dst_dir, exist_ok=True)
    with open(os.path.join(dst_dir, f"{tag}_octree_kernels.md"), "w") as g:
        g.write("\n".join(out) + "\n")
    print("\n".join(out))
    # machine-readable twin for bench.py's `octree` record (scripts/octree_bench.py hbm_traffic): short kernel name -> bytes
    import subprocess
    doc = {"source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of scripts/octree_bench.py, session {tag}", "kernels": {},
           "scene": scenes[0] if scenes else {}}
    if scenes:
        pmc_cams = int(scenes[0].get("cams", pmc_cams))
    try:
        doc["commit"] = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], cwd=os.path.dirname(os.path.abspath(__file__)),
                                                stderr=subprocess.DEVNULL).decode().strip()
    except Exception:
        pass
    for k, (calls, ms, pct) in stats.items():
        if k in rd and k in wr:
            short = k.split("pxo::")[-1].split("<")[0]
            e = {"hbm_read_bytes_per_launch": rd[k] * 2 * 1024, "hbm_write_bytes_per_launch": wr[k] * 1024, "avg_ms": ms, "launches": calls}
            if short == "grid_weight_pow2_kernel":       # one launch serves every camera of the call: the PMC passes ran --cams pmc_cams
                e["cameras_per_launch"] = pmc_cams
            if short not in doc["kernels"] or doc["kernels"][short]["avg_ms"] * doc["kernels"][short]["launches"] < ms * calls:
                doc["kernels"][short] = e              # several instantiations: keep the one that carries the time
    with open(os.path.join(dst_dir, "octree_hbm_traffic.json" if tag != "tmp" else "octree_hbm_traffic_tmp.json"), "w") as g:
        json.dump(doc, g, indent=1)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out", sys.argv[2] if len(sys.argv) > 2 else "profiles",
         sys.argv[3] if len(sys.argv) > 3 else "r01", int(sys.argv[4]) if len(sys.argv) > 4 else 2)
