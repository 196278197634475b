#!/bin/bash
# A/B of split-precision forward variants: the x3 tests (optional) + bench.py's opt_in_bf16x3_inference record per library.
set -u
export PXO_ALLOW_VARIANT=1   # these sessions select variant libraries with PXO_LIB (plenoctree_amd/_lib.py refuses it otherwise)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
R=$PWD
for sfx in ${LIBS:-main}; do
  [ "$sfx" = "main" ] && lib=$R/plenoctree_amd/libplenoctree_hip.so || lib=$R/plenoctree_amd/libplenoctree_hip$sfx.so
  [ -f "$lib" ] || { echo "missing $lib"; continue; }
  [ "${X3_TESTS:-0}" = "1" ] && { PXO_LIB=$lib timeout 600 python -m pytest tests/test_gpu_x3.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/x3test$sfx.log 2>&1; echo "pytest$sfx exit $?"; tail -3 gpurun_out/x3test$sfx.log; }
  PXO_LIB=$lib timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/x3ab$sfx.json 2> gpurun_out/x3ab$sfx.err
  echo "== $sfx exit $?"
  python - "$sfx" <<'PY'
import json, sys
try:
    d = json.load(open(f"gpurun_out/x3ab{sys.argv[1]}.json"))
    x = d["opt_in_bf16x3_inference"]
    print({k: (round(v, 2) if isinstance(v, float) else v) for k, v in x.items() if k != "note"})
except Exception as e:
    print("no result", e)
PY
done
