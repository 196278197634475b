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
if [ "${DO_PROF:-0}" = "1" ]; then
  R=$PWD; cd /tmp
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/oprof" -o obench -- python "$R/scripts/octree_bench.py" --cams 4 > "$R/gpurun_out/oprof_bench.json" 2> "$R/gpurun_out/oprof.err"
  echo "octree rocprof exit $?"
  i=0
  for ctr in FETCH_SIZE WRITE_SIZE; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d "$R/gpurun_out/opmc$i" -o pmc -- python "$R/scripts/octree_bench.py" --cams 2 > /dev/null 2> "$R/gpurun_out/opmc$i.err"
    echo "octree pmc pass $i exit $?"
  done
  cd "$R"
  find gpurun_out -name "*kernel_trace.csv" -size +20M -delete
  python scripts/summarize_octree_prof.py gpurun_out gpurun_out tmp
fi
