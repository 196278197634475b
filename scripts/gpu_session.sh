#!/bin/bash
# The ONE GPU-box session script (replaces the per-round gpu_*.sh files).  Runs the stages named in $STAGES, in order:
#
#   tests      pytest -m gpu (PYTEST_ARGS / PYTEST_K narrow it), then __graft_entry__ smoke
#   bench      python bench.py $BENCH_ARGS                     -> gpurun_out/bench.json
#   prof       rocprofv3 --kernel-trace --stats of the headline bench (+ gap analysis)   -> gpurun_out/prof
#   pmc        four rocprofv3 --pmc passes of the same command (own runs, no trace domains) -> gpurun_out/pmc{1..4}
#              (PMC_CMD="python scripts/x6_probe.py --variants x6 --steps 2": the passes of another command; PMC_TAG names the summary)
#   timeline   per-step kernel timelines at the batch sizes in $BATCHES (default 512)      -> gpurun_out/timeline_b*.txt
#   ab_trees   bench.py alternately in the trees $TREES ("_ab_base ." default), ROUNDS x per batch -> gpurun_out/ab_trees.txt
#   ab_env     run-time / variant-library A/B of ONE tree: VARIANTS="name|ENV=v ENV2=v|bench args;..." -> gpurun_out/$AB_TAG.txt
#   octree     tests/test_gpu_octree.py + scripts/octree_bench.py $OBENCH_ARGS               -> gpurun_out/octree_bench.json
#   oprof      rocprofv3 stats + FETCH_SIZE / WRITE_SIZE passes of scripts/octree_bench.py   -> gpurun_out/oprof, opmc{1,2}
#   pipeline   train -> eval -> extraction -> optimization -> evaluation through the drop-in CLIs -> gpurun_out/converge.log
#   formats    the five CLIs on the analytic scene exported in the reference's ON-DISK formats (scripts/export_scene.py): NeRF-Synthetic
#              directory (100 x 800 x 800 RGBA PNGs, --config blender) and NSVF directory (100 x 1920 x 1080, SH25 tt preset with
#              the scene's near / far), next to the same run on datasets.Synthetic            -> gpurun_out/pipeline_cli_{blender,nsvf,synthetic*}.log
#   power      clocks / power sampled while a long bench runs                               -> gpurun_out/smi.log
#   probe      scripts/contention_probe.py: a stand-in for a collective's kernel beside the step  -> gpurun_out/contention_probe.jsonl
#   tune       scripts/tune_ab.py $TUNE_ARGS: in-process A/B of pxo_set_tuning variants            -> gpurun_out/tune_ab.jsonl
#
#   gpurun --timeout 1500 -- 'STAGES="tests bench prof" bash scripts/gpu_session.sh'
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
HEAD_ARGS="--no-cpu-baseline --no-extras"
PROF_HEAD_ARGS="$HEAD_ARGS --no-kernel-events"      # under rocprofv3 the HIP-event brackets are only extra barrier packets
nproc > gpurun_out/device.txt; rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit|Max Clock" | head -8 >> gpurun_out/device.txt

brief() {   # one line per bench JSON: rays/s, ms/step, kernels
  python - "$1" "${2:-}" "${3:-}" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(sys.argv[2], "B", sys.argv[3] or d["config"]["rays_per_gpu"], round(d["value"]), "rays/s", round(d["ms_per_step"], 4), "ms/step",
          [(k["kernel"][:14], round(k["avg_ms"], 4), round(k.get("tflops", 0), 1)) for k in d["kernels"]])
except Exception as e:
    print(sys.argv[2], "B", sys.argv[3], "no result", e)
PY
}
medians() {
  python - "$1" <<'PY' | tee -a "$1"
import collections, statistics, sys
runs = collections.defaultdict(list)
for ln in open(sys.argv[1]):
    p = ln.split()
    if len(p) > 6 and p[1] == "B" and p[4] == "rays/s":
        runs[(p[0], p[2])].append(float(p[5]))
for (t, b), v in sorted(runs.items(), key=lambda kv: (int(kv[0][1]), kv[0][0])):
    print(f"median  {t:14s} B {b:>5s}: {statistics.median(v):.4f} ms/step = {int(b) / statistics.median(v) * 1e3:,.0f} rays/s  (n={len(v)}, min {min(v):.4f})")
PY
}

for stage in ${STAGES:-tests bench}; do
  echo "=== stage $stage"
  case $stage in
  tests)
    rm -f gpurun_out/fullsize_parity.jsonl gpurun_out/trained_state_parity.jsonl
    timeout ${TEST_TIMEOUT:-1800} python -m pytest tests -m gpu --durations=15 -q --tb=short -p no:cacheprovider ${PYTEST_K:+-k "$PYTEST_K"} ${PYTEST_ARGS:-} > gpurun_out/pytest_gpu.log 2>&1
    echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
    tail -${TEST_TAIL:-40} gpurun_out/pytest_gpu.log
    cat gpurun_out/fullsize_parity.jsonl gpurun_out/trained_psnr.json gpurun_out/trained_psnr_twin512.json gpurun_out/trained_state_parity.jsonl 2>/dev/null
    timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log
    tail -3 gpurun_out/smoke.log ;;
  bench)
    t0=$SECONDS
    timeout ${BENCH_TIMEOUT:-1200} python bench.py ${BENCH_ARGS:-} > gpurun_out/bench.json 2> gpurun_out/bench.err
    echo "bench exit $? wall $((SECONDS - t0)) s" | tee gpurun_out/bench_wall.txt; cat gpurun_out/bench.json; tail -3 gpurun_out/bench.err ;;
  prof)
    cd /tmp
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof" -o bench -- python "$R/bench.py" --steps 10 --warmup 3 $PROF_HEAD_ARGS ${PROF_ARGS:-} > "$R/gpurun_out/prof_bench.json" 2> "$R/gpurun_out/prof.err"
    echo "rocprof exit $?"; cd "$R"
    f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -16 "$f"
    f=$(find gpurun_out/prof -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python scripts/gap_analysis.py "$f" | tee gpurun_out/gaps.txt
    find gpurun_out -name "*.csv" -size +30M -delete ;;
  pmc)
    cd /tmp; i=0
    for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"; do
      i=$((i+1))
      if [ -n "${PMC_CMD:-}" ]; then
        ( cd "$R" && timeout 600 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d "$R/gpurun_out/pmc$i" -o pmc -- $PMC_CMD > /dev/null 2> "$R/gpurun_out/pmc$i.err" )
      else
        timeout 600 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d "$R/gpurun_out/pmc$i" -o pmc -- python "$R/bench.py" --steps 2 --warmup 1 $PROF_HEAD_ARGS > /dev/null 2> "$R/gpurun_out/pmc$i.err"
      fi
      echo "pmc pass $i exit $?"
    done
    cd "$R"; find gpurun_out -name "*.csv" -size +30M -delete
    # a summary on the box (the full CSVs can exceed what is merged back); hbm_traffic.json of a custom command is not the bench's
    mkdir -p gpurun_out/pmc_summary && python scripts/summarize_prof.py gpurun_out gpurun_out/pmc_summary ${PMC_TAG:-pmc} > /dev/null
    cat gpurun_out/pmc_summary/${PMC_TAG:-pmc}_pmc.md ;;
  timeline)
    for B in ${BATCHES:-512}; do
      timeout 300 python bench.py --batch $B --steps 40 --warmup 5 $HEAD_ARGS ${TL_ARGS:-} > gpurun_out/bench_b$B.json 2> gpurun_out/bench_b$B.err
      echo "bench B=$B exit $?"; brief gpurun_out/bench_b$B.json timeline $B
      cd /tmp
      timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$R/gpurun_out/trace_b$B" -o t -- python "$R/bench.py" --batch $B --steps 12 --warmup 3 $PROF_HEAD_ARGS ${TL_ARGS:-} > /dev/null 2> "$R/gpurun_out/trace_b$B.err"
      echo "rocprof B=$B exit $?"; cd "$R"
      f=$(find gpurun_out/trace_b$B -name "*kernel_trace.csv" | head -1)
      [ -n "$f" ] && python scripts/step_timeline.py "$f" 4 > gpurun_out/timeline_b$B.txt && head -${TL_HEAD:-50} gpurun_out/timeline_b$B.txt
      find gpurun_out/trace_b$B -name "*.csv" -size +20M -delete
    done ;;
  ab_trees)
    : > gpurun_out/ab_trees.txt
    TREES_=${TREES:-_ab_base .}
    for B in ${BATCHES:-512}; do for r in $(seq 1 ${ROUNDS:-3}); do for t in $TREES_; do
      ( cd "$R/$t" && timeout 300 python bench.py --batch $B --steps ${AB_STEPS:-60} --warmup 5 $HEAD_ARGS ${AB_ARGS:-} 2> /dev/null ) > gpurun_out/ab_run.json
      brief gpurun_out/ab_run.json "$t" $B | tee -a gpurun_out/ab_trees.txt
    done; done; done
    medians gpurun_out/ab_trees.txt ;;
  ab_env)
    OUT=gpurun_out/${AB_TAG:-ab_env}.txt; : > $OUT
    IFS=';' read -ra VARS <<< "${VARIANTS:-base||}"
    for B in ${BATCHES:-512}; do for r in $(seq 1 ${ROUNDS:-3}); do for v in "${VARS[@]}"; do
      IFS='|' read -r name envs args <<< "$v"
      ( [ -n "$envs" ] && export $envs; timeout 300 python bench.py --batch $B --steps ${AB_STEPS:-60} --warmup 5 $HEAD_ARGS $args 2> gpurun_out/ab_run.err ) > gpurun_out/ab_run.json
      brief gpurun_out/ab_run.json "$name" $B | tee -a $OUT
    done; done; done
    medians $OUT ;;
  octree)
    timeout ${TEST_TIMEOUT:-600} python -m pytest tests/test_gpu_octree.py -m gpu --durations=8 -q --tb=short -p no:cacheprovider -x > gpurun_out/pytest_octree.log 2>&1
    echo "pytest exit $?" >> gpurun_out/pytest_octree.log; tail -12 gpurun_out/pytest_octree.log
    timeout 300 python scripts/octree_bench.py ${OBENCH_ARGS:-} > gpurun_out/octree_bench.json 2> gpurun_out/octree_bench.err
    echo "octree_bench exit $?"; cat gpurun_out/octree_bench.json; tail -5 gpurun_out/octree_bench.err ;;
  oprof)
    cd /tmp
    # stats AND both counter passes on the scene of bench.py's `octree` record (--cams ${OCAMS:-4}): the tree depends on the camera set
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/oprof" -o obench -- python "$R/scripts/octree_bench.py" --cams ${OCAMS:-4} > "$R/gpurun_out/oprof_bench.json" 2> "$R/gpurun_out/oprof.err"
    echo "octree rocprof exit $?"; i=0
    for ctr in FETCH_SIZE WRITE_SIZE; do
      i=$((i+1))
      timeout 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d "$R/gpurun_out/opmc$i" -o pmc -- python "$R/scripts/octree_bench.py" --cams ${OCAMS:-4} --no-roofline > "$R/gpurun_out/opmc$i.json" 2> "$R/gpurun_out/opmc$i.err"
      echo "octree pmc pass $i exit $?"
    done
    cd "$R"; find gpurun_out -name "*kernel_trace.csv" -size +20M -delete
    python scripts/summarize_octree_prof.py gpurun_out gpurun_out ${OPROF_TAG:-tmp} ;;
  pipeline)
    mkdir -p /tmp/pxo_conv
    printf 'dataset: synthetic\nfactor: 0\nnum_coarse_samples: 64\nnum_fine_samples: 128\nuse_viewdirs: false\nwhite_bkgd: true\nbatch_size: 4096\nsh_deg: 3\nrandomized: true\nmax_steps: %s\nprint_every: 250\nsave_every: %s\nrender_every: 1000\nchunk: 8192\n' ${PIPE_STEPS:-3000} ${PIPE_STEPS:-3000} > /tmp/pxo_conv/cfg.yaml
    [ -n "${PIPE_YAML_EXTRA:-}" ] && printf "${PIPE_YAML_EXTRA}\n" >> /tmp/pxo_conv/cfg.yaml      # e.g. PIPE_YAML_EXTRA='mlp_precision: bf16x6'
    C="--train_dir /tmp/pxo_conv --config /tmp/pxo_conv/cfg.yaml"
    timeout 400 python -m plenoctree_amd.nerf_sh.train $C > gpurun_out/converge.log 2>&1; echo "train exit $?"
    timeout 200 python -m plenoctree_amd.nerf_sh.eval $C --approx_eval_skip 50 --save_output false >> gpurun_out/converge.log 2>&1; echo "eval exit $?"
    timeout 300 python -m plenoctree_amd.octree.extraction $C --init_grid_depth 8 --output /tmp/pxo_conv/tree.npz >> gpurun_out/converge.log 2>&1; echo "extraction exit $?"
    timeout 300 python -m plenoctree_amd.octree.optimization $C --input /tmp/pxo_conv/tree.npz --output /tmp/pxo_conv/tree_opt.npz --num_epochs ${OPT_EPOCHS:-4} --val_interval 2 >> gpurun_out/converge.log 2>&1; echo "optimization exit $?"
    timeout 200 python -m plenoctree_amd.octree.evaluation $C --input /tmp/pxo_conv/tree_opt.npz >> gpurun_out/converge.log 2>&1; echo "evaluation exit $?"
    grep -v amdgpu.ids gpurun_out/converge.log | tail -45 ;;
  formats)
    D=/tmp/pxo_formats; rm -rf $D; mkdir -p $D
    STEPS=${PIPE_STEPS:-3000}
    timeout 600 python scripts/export_scene.py blender $D/scene_blender --size 800 800 --views 100 8 8 > gpurun_out/export_scene.log 2>&1; echo "export blender exit $?"
    timeout 900 python scripts/export_scene.py nsvf $D/scene_nsvf --size 1080 1920 --views 100 8 8 >> gpurun_out/export_scene.log 2>&1; echo "export nsvf exit $?"
    du -sh $D/scene_blender $D/scene_nsvf | tee -a gpurun_out/export_scene.log
    common='image_batching: false\nfactor: 0\nnum_coarse_samples: 64\nnum_fine_samples: 128\nuse_viewdirs: false\nwhite_bkgd: true\nbatch_size: 4096\nrandomized: true\nnear: 2.0\nfar: 6.0\nprint_every: 500\nrender_every: 100000\nchunk: 8192\n'
    for run in blender synthetic_as_blender nsvf synthetic_as_nsvf; do
      T=$D/train_$run; mkdir -p $T
      case $run in
        blender) printf "dataset: blender\nsh_deg: 3\n$common" > $T/cfg.yaml; DD="--data_dir $D/scene_blender" ;;
        synthetic_as_blender) printf "dataset: synthetic\nsynthetic_8bit: true\nsynthetic_views: [100, 8]\nsh_deg: 3\n$common" > $T/cfg.yaml; DD="" ;;
        nsvf) printf "dataset: nsvf\nsh_deg: 4\nsparsity_radius: 1.5\nsparsity_length: 0.05\n$common" > $T/cfg.yaml; DD="--data_dir $D/scene_nsvf" ;;
        synthetic_as_nsvf) printf "dataset: synthetic\nsynthetic_8bit: true\nsynthetic_views: [100, 8]\nsynthetic_hw: [1080, 1920]\nsh_deg: 4\nsparsity_radius: 1.5\nsparsity_length: 0.05\n$common" > $T/cfg.yaml; DD="" ;;
      esac
      printf "max_steps: $STEPS\nsave_every: $STEPS\n" >> $T/cfg.yaml
      C="--train_dir $T --config $T/cfg.yaml $DD"
      L=gpurun_out/pipeline_cli_$run.log; : > $L
      ( echo "### cfg.yaml"; cat $T/cfg.yaml; [ -n "$DD" ] && echo "### data_dir: $(ls ${DD#--data_dir } | head -8 | tr '\n' ' ')" ) >> $L
      timeout 900 python -m plenoctree_amd.nerf_sh.train $C >> $L 2>&1; echo "$run train exit $?"
      timeout 600 python -m plenoctree_amd.nerf_sh.eval $C --save_output false >> $L 2>&1; echo "$run eval exit $?"
      timeout 600 python -m plenoctree_amd.octree.extraction $C --init_grid_depth 8 --output $T/tree.npz >> $L 2>&1; echo "$run extraction exit $?"
      timeout 600 python -m plenoctree_amd.octree.optimization $C --input $T/tree.npz --output $T/tree_opt.npz --num_epochs ${OPT_EPOCHS:-2} --val_interval 1 >> $L 2>&1; echo "$run optimization exit $?"
      timeout 600 python -m plenoctree_amd.octree.evaluation $C --input $T/tree_opt.npz >> $L 2>&1; echo "$run evaluation exit $?"
      grep -v amdgpu.ids $L | grep -iE "psnr|rays/s|exit|ssim|error|Traceback" | tail -25
    done
    # the PSNR lines of the on-disk runs next to those of the same scene through datasets.Synthetic: they must be the same lines
    for pair in "blender synthetic_as_blender" "nsvf synthetic_as_nsvf"; do
      set -- $pair
      echo "== $1 vs $2 (PSNR lines that differ; none = identical)"
      diff <(grep -iE "psnr" gpurun_out/pipeline_cli_$1.log | grep -v "rays/s") <(grep -iE "psnr" gpurun_out/pipeline_cli_$2.log | grep -v "rays/s") | head -20
    done ;;
  power)
    ( for i in $(seq 1 40); do rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|Power|mclk|Temperature \(Sensor (edge|junction)" | tr '\n' ' ' ; echo; sleep 0.5; done ) > gpurun_out/smi.log &
    timeout 120 python bench.py --steps 400 --warmup 3 $HEAD_ARGS > gpurun_out/bench_long.json 2> gpurun_out/bench_long.err
    wait; brief gpurun_out/bench_long.json power; sed -n '1p;8p;16p;24p;32p' gpurun_out/smi.log | cut -c1-400 ;;
  probe)
    timeout 300 python scripts/contention_probe.py > gpurun_out/contention_probe.jsonl 2> gpurun_out/contention_probe.err
    echo "probe exit $?"; cut -c1-170 gpurun_out/contention_probe.jsonl ;;
  tune)
    timeout 400 python scripts/tune_ab.py ${TUNE_ARGS:---batches 512,4096 static:tile_sched=0 counter:tile_sched=1} > gpurun_out/tune_ab.jsonl 2> gpurun_out/tune_ab.err
    echo "tune exit $?"; cut -c1-200 gpurun_out/tune_ab.jsonl ;;
  *) echo "unknown stage $stage" ;;
  esac
done
