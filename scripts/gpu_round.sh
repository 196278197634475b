#!/bin/bash
# One GPU-box session: parity tests, smoke, short bench, rocprofv3 kernel stats.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
rocm-smi --showproductname 2>/dev/null | head -8 > gpurun_out/device.txt
nproc >> gpurun_out/device.txt
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -40 gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log
tail -3 gpurun_out/smoke.log
timeout 600 python bench.py --steps ${BENCH_STEPS:-10} --warmup 3 > gpurun_out/bench.json 2> gpurun_out/bench.err
echo "bench exit $?"; cat gpurun_out/bench.json; tail -5 gpurun_out/bench.err
if [ "${DO_PROF:-1}" = "1" ]; then
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof" -o bench -- python "$OLDPWD/bench.py" --steps 5 --warmup 2 --no-cpu-baseline > "$OLDPWD/gpurun_out/prof_bench.json" 2> "$OLDPWD/gpurun_out/prof.err")
  echo "rocprof exit $?"
  find gpurun_out/prof -name "*stats*" | head; 
  f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -25 "$f"
  # keep the merge-back small: drop the raw trace, keep stats
  find gpurun_out/prof -name "*kernel_trace.csv" -size +20M -delete
fi
