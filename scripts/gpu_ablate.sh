#!/bin/bash
# Timing-only A/B of kernel variants (PXO_LIB): prints rays/s and per-kernel ms/TFLOPs.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; R=$PWD
for v in "" _abl1 _abl2 _abl3 $EXTRA_VARIANTS; do
  lib=$R/plenoctree_amd/libplenoctree_hip$v.so
  [ -f "$lib" ] || continue
  PXO_LIB=$lib timeout 120 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/bench$v.json 2> gpurun_out/bench$v.err
  python -c "import json,sys;d=json.load(open('gpurun_out/bench$v.json'));print('variant[$v]',round(d['value']),[(k['kernel'][:12],round(k['avg_ms'],3),round(k.get('tflops',0),1)) for k in d['kernels']])" || tail -3 gpurun_out/bench$v.err
done
