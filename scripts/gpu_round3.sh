#!/bin/bash
# Round-3 profile session: rocprofv3 kernel stats + 4 PMC passes at the headline shape, the per-step timeline of the
# 512-rays-per-GPU strong-scaling shape, and the cycle-stamp trace of the forward kernel (scripts/trace_mlp.py).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
DO_TESTS=0 DO_BENCH=0 DO_PROF=1 DO_PMC=1 bash scripts/gpu_round2.sh 2>&1 | tail -40
BATCHES='512' bash scripts/gpu_r3_probe.sh 2>&1 | tail -8
cp gpurun_out/timeline_b512.txt gpurun_out/r03_timeline_b512.txt 2>/dev/null
if [ -f plenoctree_amd/libplenoctree_hip_trace.so ]; then
  PXO_ALLOW_VARIANT=1 PXO_LIB=$R/plenoctree_amd/libplenoctree_hip_trace.so timeout 300 python scripts/trace_mlp.py run > gpurun_out/r03_mlp_fwd_phases.txt 2>&1
  tail -12 gpurun_out/r03_mlp_fwd_phases.txt
fi
find gpurun_out -name "*.csv" -size +20M -delete
