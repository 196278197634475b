#!/bin/bash
# Sample clocks/power while the bench runs (is the step power/DVFS limited?).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
( for i in $(seq 1 40); do rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|Power|mclk|Temperature \(Sensor (edge|junction)" | tr '\n' ' ' ; echo; sleep 0.5; done ) > gpurun_out/smi.log &
timeout 120 python bench.py --steps 400 --warmup 3 --no-cpu-baseline > gpurun_out/bench_long.json 2> gpurun_out/bench_long.err
wait
python -c "import json;d=json.load(open('gpurun_out/bench_long.json'));print(d['value'],d['ms_per_step'])"
sed -n '1p;8p;16p;24p;32p' gpurun_out/smi.log | cut -c1-400
