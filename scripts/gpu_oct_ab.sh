#!/bin/bash
# A/B of octree kernel variants: scripts/octree_bench.py once per library in $LIBS (suffixes of libplenoctree_hip<suffix>.so).
set -u
export PXO_ALLOW_VARIANT=1   # these sessions select variant libraries with PXO_LIB (plenoctree_amd/_lib.py refuses it otherwise)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
R=$PWD
[ -n "${PYTEST_ARGS:-}" ] && { timeout 600 python -m pytest ${PYTEST_ARGS} ${PYTEST_K:+-k "$PYTEST_K"} -q --tb=short -p no:cacheprovider > gpurun_out/pytest_ab.log 2>&1; echo "pytest exit $?"; tail -15 gpurun_out/pytest_ab.log; }
for sfx in ${LIBS:-main}; do
  [ "$sfx" = "main" ] && lib=$R/plenoctree_amd/libplenoctree_hip.so || lib=$R/plenoctree_amd/libplenoctree_hip$sfx.so
  [ -f "$lib" ] || { echo "missing $lib"; continue; }
  PXO_LIB=$lib timeout 300 python scripts/octree_bench.py --cams ${CAMS:-4} ${OB_ARGS:-} > gpurun_out/oab$sfx.json 2> gpurun_out/oab$sfx.err
  echo "== $sfx exit $?"
  python - "$sfx" <<'PY'
import json, sys
try:
    d = json.load(open(f"gpurun_out/oab{sys.argv[1]}.json"))
    print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in d.items() if "ms" in k or "samples" in k})
except Exception as e:
    print("no result", e)
PY
done
