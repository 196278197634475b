#!/bin/bash
# In-session A/B of two source trees (e.g. _ab_base = the previous round's HEAD vs the working tree): bench.py (headline
# only) run alternately in each tree, ROUNDS times per batch size; prints every run and the per-tree median.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
mkdir -p gpurun_out
TREES=${TREES:-"_ab_base ."}
: > gpurun_out/ab_trees.txt
for B in ${BATCHES:-512}; do
  for r in $(seq 1 ${ROUNDS:-3}); do
    for t in $TREES; do
      ( cd "$R/$t" && timeout 300 python bench.py --batch $B --steps ${AB_STEPS:-60} --warmup 5 --no-cpu-baseline --no-extras ${AB_ARGS:-} 2> /dev/null ) > gpurun_out/ab_run.json
      python - "$t" "$B" <<'PY' | tee -a gpurun_out/ab_trees.txt
import json, sys
try:
    d = json.load(open("gpurun_out/ab_run.json"))
    print(sys.argv[1], "B", sys.argv[2], round(d["value"]), "rays/s", round(d["ms_per_step"], 4), "ms/step", [(k["kernel"][:14], round(k["avg_ms"], 4)) for k in d["kernels"]])
except Exception as e:
    print(sys.argv[1], "B", sys.argv[2], "no result", e)
PY
    done
  done
done
python - <<'PY' | tee -a gpurun_out/ab_trees.txt
import collections, statistics
runs = collections.defaultdict(list)
for ln in open("gpurun_out/ab_trees.txt"):
    p = ln.split()
    if len(p) > 6 and p[1] == "B" and p[4] == "rays/s":
        runs[(p[0], p[2])].append(float(p[5]))
for (t, b), v in sorted(runs.items(), key=lambda kv: (int(kv[0][1]), kv[0][0])):
    print(f"median  tree {t:10s} B {b:>5s}: {statistics.median(v):.4f} ms/step = {int(b) / statistics.median(v) * 1e3:,.0f} rays/s  (n={len(v)}, min {min(v):.4f})")
PY
