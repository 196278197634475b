#!/bin/bash
# Round-3 closing session: all GPU tests + smoke + default bench, then the profile passes (gpu_round3.sh) and the
# PlenOctree-side kernels with their PMC passes (gpu_octree.sh).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
DO_TESTS=1 DO_BENCH=1 DO_PROF=0 bash scripts/gpu_round2.sh 2>&1 | tail -25
bash scripts/gpu_round3.sh 2>&1 | tail -30
TEST_TIMEOUT=1 DO_BENCH=1 DO_PROF=1 BENCH_ARGS="--cams 8" bash scripts/gpu_octree.sh 2>&1 | tail -30
