#!/bin/bash
# GPU session for the PlenOctree side: parity tests + kernel timings at the reference's sizes.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout ${TEST_TIMEOUT:-420} python -m pytest tests/test_gpu_octree.py -m gpu --durations=8 -q --tb=short -p no:cacheprovider -x > gpurun_out/pytest_octree.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_octree.log
tail -40 gpurun_out/pytest_octree.log
if [ "${DO_BENCH:-1}" = "1" ]; then
  timeout 300 python scripts/octree_bench.py ${BENCH_ARGS:-} > gpurun_out/octree_bench.json 2> gpurun_out/octree_bench.err
  echo "octree_bench exit $?"; cat gpurun_out/octree_bench.json; tail -5 gpurun_out/octree_bench.err
fi
