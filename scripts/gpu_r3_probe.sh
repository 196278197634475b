#!/bin/bash
# Round-3 probe session: targeted tests, then per-step kernel timelines at small per-GPU batches (the strong-scaling shape).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
if [ -n "${PYTEST_K:-}" ]; then
  rm -f gpurun_out/fullsize_parity.jsonl
  timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "$PYTEST_K" > gpurun_out/pytest_k.log 2>&1
  echo "pytest exit $?"; tail -15 gpurun_out/pytest_k.log; cat gpurun_out/fullsize_parity.jsonl 2>/dev/null
fi
for B in ${BATCHES:-512}; do
  timeout 300 python bench.py --batch $B --steps 40 --warmup 5 --no-cpu-baseline --no-extras ${BENCH_ARGS:-} > gpurun_out/bench_b$B.json 2> gpurun_out/bench_b$B.err
  echo "bench B=$B exit $?"; python -c "
import json; d = json.load(open('gpurun_out/bench_b$B.json'))
print(round(d['value']), 'rays/s', round(d['ms_per_step'], 4), 'ms/step', [(k['kernel'][:18], k['launches'], round(k['avg_ms'], 4), round(k.get('tflops', 0), 1)) for k in d['kernels']])"
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$R/gpurun_out/trace_b$B" -o t -- python "$R/bench.py" --batch $B --steps 12 --warmup 3 --no-cpu-baseline --no-extras ${BENCH_ARGS:-} > /dev/null 2> "$R/gpurun_out/trace_b$B.err"
  echo "rocprof B=$B exit $?"
  cd "$R"
  f=$(find gpurun_out/trace_b$B -name "*kernel_trace.csv" | head -1)
  [ -n "$f" ] && python scripts/step_timeline.py "$f" 4 > gpurun_out/timeline_b$B.txt && head -45 gpurun_out/timeline_b$B.txt
  find gpurun_out/trace_b$B -name "*.csv" -size +20M -delete
done
