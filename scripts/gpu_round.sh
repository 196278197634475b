#!/bin/bash
# One GPU-box session: parity tests, smoke, bench (A/B of kernel variants), rocprofv3 stats + PMC.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
nproc > gpurun_out/device.txt; rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit|Max Clock" | head -8 >> gpurun_out/device.txt
if [ "${DO_TESTS:-1}" = "1" ]; then
  timeout 420 python -m pytest tests -m gpu --durations=5 -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
  echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
  tail -25 gpurun_out/pytest_gpu.log
  timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log
  tail -2 gpurun_out/smoke.log
fi
timeout 600 python bench.py --steps ${BENCH_STEPS:-10} --warmup 3 > gpurun_out/bench.json 2> gpurun_out/bench.err
echo "bench exit $?"; cat gpurun_out/bench.json; tail -3 gpurun_out/bench.err
timeout 200 python bench.py --preset tt --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_tt.json 2> gpurun_out/bench_tt.err
echo "bench(tt SH25) exit $?"; python -c "import json;d=json.load(open('gpurun_out/bench_tt.json'));print(round(d['value']),d['ms_per_step'],[(k['kernel'][:14],round(k['avg_ms'],3),round(k.get('tflops',0),1)) for k in d['kernels']])"
if [ -f plenoctree_amd/libplenoctree_hip_g0.so ]; then
  PXO_LIB=$R/plenoctree_amd/libplenoctree_hip_g0.so timeout 300 python bench.py --steps ${BENCH_STEPS:-10} --warmup 3 --no-cpu-baseline > gpurun_out/bench_g0.json 2> gpurun_out/bench_g0.err
  echo "bench(g0 variant) exit $?"; python -c "import json;d=json.load(open('gpurun_out/bench_g0.json'));print(d['value'],[(k['kernel'],round(k['avg_ms'],3),round(k.get('tflops',0),1)) for k in d['kernels']])"
fi
if [ "${DO_PROF:-1}" = "1" ]; then
  cd /tmp
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof" -o bench -- python "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline > "$R/gpurun_out/prof_bench.json" 2> "$R/gpurun_out/prof.err"
  echo "rocprof exit $?"
  i=0
  [ "${DO_PMC:-1}" = "1" ] && for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout 600 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d "$R/gpurun_out/pmc$i" -o pmc -- python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> "$R/gpurun_out/pmc$i.err"
    echo "pmc pass $i exit $?"
  done
  cd "$R"
  find gpurun_out/prof gpurun_out/pmc* -type f | head -30
  f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f"
  f=$(find gpurun_out/prof -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python scripts/gap_analysis.py "$f" | tee gpurun_out/gaps.txt
  find gpurun_out -name "*.csv" -size +30M -delete
fi
