#!/bin/bash
# A/B of octree-kernel launch switches: scripts/octree_bench.py once per entry of $ENVS ("NAME=V,NAME=V" per entry, "-" = none).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
if [ -n "${PYTEST_ENVS:-}" ]; then
  for e in $PYTEST_ENVS; do
    [ "$e" = "-" ] && ev="" || ev=$(echo "$e" | tr ',' ' ')
    env $ev timeout 600 python -m pytest tests/test_gpu_octree.py -m gpu -q --tb=short -p no:cacheprovider ${PYTEST_K:+-k "$PYTEST_K"} > gpurun_out/pytest_oct_$e.log 2>&1
    echo "pytest [$e] exit $?"; tail -3 gpurun_out/pytest_oct_$e.log
  done
fi
i=0
for e in ${ENVS:--}; do
  [ "$e" = "-" ] && ev="" || ev=$(echo "$e" | tr ',' ' ')
  ( cd ${BENCH_ROOT:-.} && env $ev timeout 300 python scripts/octree_bench.py --cams ${CAMS:-4} ${OB_ARGS:-} ) > gpurun_out/oenv_${TAG:-}$i.json 2> gpurun_out/oenv_${TAG:-}$i.err
  echo "== [$e] exit $?"
  python - "$i" "${TAG:-}" <<'PY'
import json, sys
try:
    d = json.load(open(f"gpurun_out/oenv_{sys.argv[2]}{sys.argv[1]}.json"))
    print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in d.items() if ("ms" in k or "sum" in k or "fraction" in k) and "tree" not in k and "sample" not in k})
except Exception as ex:
    print("no result", ex)
PY
  i=$((i+1))
done
