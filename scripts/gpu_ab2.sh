#!/bin/bash
# Tests (optional) + bench A/B at several batch sizes: for each B in $BATCHES, bench.py once per library in $LIBS.
set -u
export PXO_ALLOW_VARIANT=1   # these sessions select variant libraries with PXO_LIB (plenoctree_amd/_lib.py refuses it otherwise)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
[ -n "${PYTEST_ARGS:-}" ] && { timeout 900 python -m pytest ${PYTEST_ARGS} ${PYTEST_K:+-k "$PYTEST_K"} -q --tb=short -p no:cacheprovider > gpurun_out/pytest_ab.log 2>&1; echo "pytest exit $?"; tail -8 gpurun_out/pytest_ab.log; }
for B in ${BATCHES:-4096}; do
  echo "#### batch $B"
  AB_ARGS="--batch $B ${EXTRA_ARGS:-}" bash scripts/gpu_ab.sh
done
