#!/bin/bash
# A/B of kernel variants in one GPU session: bench.py (headline only) once per library in $LIBS (suffixes of
# plenoctree_amd/libplenoctree_hip<suffix>.so; "" = the default build).
set -u
export PXO_ALLOW_VARIANT=1   # these sessions select variant libraries with PXO_LIB (plenoctree_amd/_lib.py refuses it otherwise)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
R=$PWD
LIBS=${LIBS:-main _bd2 _ws0 _nohalf}
for sfx in $LIBS; do
  [ "$sfx" = "main" ] && lib=$R/plenoctree_amd/libplenoctree_hip.so || lib=$R/plenoctree_amd/libplenoctree_hip$sfx.so
  [ -f "$lib" ] || { echo "missing $lib"; continue; }
  PXO_LIB=$lib timeout 300 python bench.py --steps ${AB_STEPS:-40} --warmup 5 --no-cpu-baseline --no-extras ${AB_ARGS:-} > gpurun_out/ab$sfx.json 2> gpurun_out/ab$sfx.err
  echo "== $sfx exit $?"
  python - "$sfx" <<'PY'
import json, sys
try:
    d = json.load(open(f"gpurun_out/ab{sys.argv[1]}.json"))
    print(round(d["value"]), "rays/s", round(d["ms_per_step"], 3), "ms/step", [(k["kernel"][:18], k["launches"], round(k["avg_ms"], 4), round(k.get("tflops", 0), 1)) for k in d["kernels"]])
except Exception as e:
    print("no result", e)
PY
done
