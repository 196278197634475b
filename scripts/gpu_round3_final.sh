#!/bin/bash
# Round-3 final session on the NeRF side: all GPU tests + smoke + default bench + rocprofv3 stats and PMC passes, the
# 512-rays-per-GPU timeline and the forward kernel's cycle-stamp trace.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
DO_TESTS=1 DO_BENCH=1 DO_PROF=1 DO_PMC=1 bash scripts/gpu_round2.sh 2>&1 | tail -45
R=$PWD
BATCHES='512' bash scripts/gpu_r3_probe.sh 2>&1 | tail -6
cp gpurun_out/timeline_b512.txt gpurun_out/r03_timeline_b512.txt 2>/dev/null
if [ -f plenoctree_amd/libplenoctree_hip_trace.so ]; then
  PXO_ALLOW_VARIANT=1 PXO_LIB=$R/plenoctree_amd/libplenoctree_hip_trace.so timeout 300 python scripts/trace_mlp.py run > gpurun_out/r03_mlp_fwd_phases.txt 2>&1
  tail -8 gpurun_out/r03_mlp_fwd_phases.txt
fi
find gpurun_out -name "*.csv" -size +20M -delete
