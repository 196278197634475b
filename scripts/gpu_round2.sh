#!/bin/bash
# Round-2 GPU-box session: parity tests (incl. the BASELINE-size oracle comparisons), smoke, bench with the extra records.
# DO_TESTS / DO_BENCH / DO_PROF / DO_PMC = 0|1
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
nproc > gpurun_out/device.txt; rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit|Max Clock" | head -8 >> gpurun_out/device.txt
if [ "${DO_TESTS:-1}" = "1" ]; then
  rm -f gpurun_out/fullsize_parity.jsonl
  timeout ${TEST_TIMEOUT:-1500} python -m pytest tests -m gpu --durations=15 -q --tb=short -p no:cacheprovider ${PYTEST_ARGS:-} > gpurun_out/pytest_gpu.log 2>&1
  echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
  tail -40 gpurun_out/pytest_gpu.log
  cat gpurun_out/fullsize_parity.jsonl gpurun_out/trained_psnr.json 2>/dev/null
  timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log
  tail -3 gpurun_out/smoke.log
fi
if [ "${DO_BENCH:-1}" = "1" ]; then
  timeout 900 python bench.py ${BENCH_ARGS:-} > gpurun_out/bench.json 2> gpurun_out/bench.err
  echo "bench exit $?"; cat gpurun_out/bench.json; tail -3 gpurun_out/bench.err
fi
if [ "${DO_PROF:-0}" = "1" ]; then
  cd /tmp
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof" -o bench -- python "$R/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-extras ${PROF_ARGS:-} > "$R/gpurun_out/prof_bench.json" 2> "$R/gpurun_out/prof.err"
  echo "rocprof exit $?"
  i=0
  [ "${DO_PMC:-0}" = "1" ] && for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout 600 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d "$R/gpurun_out/pmc$i" -o pmc -- python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> "$R/gpurun_out/pmc$i.err"
    echo "pmc pass $i exit $?"
  done
  cd "$R"
  f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -16 "$f"
  f=$(find gpurun_out/prof -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python scripts/gap_analysis.py "$f" | tee gpurun_out/gaps.txt
  find gpurun_out -name "*.csv" -size +30M -delete
fi
