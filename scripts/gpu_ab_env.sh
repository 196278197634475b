#!/bin/bash
# In-session A/B of run-time variants of ONE tree: VARIANTS is a ';'-separated list of "name|ENV=val ENV2=val|bench args";
# every variant runs once per round, interleaved, ROUNDS rounds per batch size; prints each run and per-variant medians.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=gpurun_out/${AB_TAG:-ab_env}.txt
: > $OUT
IFS=';' read -ra VARS <<< "${VARIANTS:-base||}"
for B in ${BATCHES:-512}; do
  for r in $(seq 1 ${ROUNDS:-3}); do
    for v in "${VARS[@]}"; do
      IFS='|' read -r name envs args <<< "$v"
      ( [ -n "$envs" ] && export $envs; timeout 300 python bench.py --batch $B --steps ${AB_STEPS:-60} --warmup 5 --no-cpu-baseline --no-extras $args 2> gpurun_out/ab_run.err ) > gpurun_out/ab_run.json
      python - "$name" "$B" <<'PY' | tee -a $OUT
import json, sys
try:
    d = json.load(open("gpurun_out/ab_run.json"))
    print(sys.argv[1], "B", sys.argv[2], round(d["value"]), "rays/s", round(d["ms_per_step"], 4), "ms/step", [(k["kernel"][:14], round(k["avg_ms"], 4)) for k in d["kernels"]])
except Exception as e:
    print(sys.argv[1], "B", sys.argv[2], "no result", e, open("gpurun_out/ab_run.err").read()[-300:])
PY
    done
  done
done
python - $OUT <<'PY' | tee -a $OUT
import collections, statistics, sys
runs = collections.defaultdict(list)
for ln in open(sys.argv[1]):
    p = ln.split()
    if len(p) > 6 and p[1] == "B" and p[4] == "rays/s":
        runs[(p[0], p[2])].append(float(p[5]))
for (t, b), v in sorted(runs.items(), key=lambda kv: (int(kv[0][1]), kv[0][0])):
    print(f"median  {t:14s} B {b:>5s}: {statistics.median(v):.4f} ms/step = {int(b) / statistics.median(v) * 1e3:,.0f} rays/s  (n={len(v)}, min {min(v):.4f})")
PY
