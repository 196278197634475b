#!/bin/bash
# End-to-end convergence demo through the drop-in CLI on the synthetic scene.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out /tmp/pxo_conv
cat > /tmp/pxo_conv/cfg.yaml <<'EOC'
dataset: synthetic
factor: 0
num_coarse_samples: 64
num_fine_samples: 128
use_viewdirs: false
white_bkgd: true
batch_size: 4096
sh_deg: 3
randomized: true
max_steps: 3000
print_every: 250
save_every: 3000
render_every: 1000
chunk: 8192
EOC
timeout 400 python -m plenoctree_amd.nerf_sh.train --train_dir /tmp/pxo_conv --config /tmp/pxo_conv/cfg.yaml > gpurun_out/converge.log 2>&1
echo "train exit $?"
timeout 200 python -m plenoctree_amd.nerf_sh.eval --train_dir /tmp/pxo_conv --config /tmp/pxo_conv/cfg.yaml --approx_eval_skip 50 --save_output false >> gpurun_out/converge.log 2>&1
echo "eval exit $?"
timeout 300 python -m plenoctree_amd.octree.extraction --train_dir /tmp/pxo_conv --config /tmp/pxo_conv/cfg.yaml --init_grid_depth 8 --output /tmp/pxo_conv/tree.npz >> gpurun_out/converge.log 2>&1
echo "extraction exit $?"
timeout 300 python -m plenoctree_amd.octree.optimization --train_dir /tmp/pxo_conv --config /tmp/pxo_conv/cfg.yaml --input /tmp/pxo_conv/tree.npz --output /tmp/pxo_conv/tree_opt.npz --num_epochs ${OPT_EPOCHS:-4} --val_interval 2 >> gpurun_out/converge.log 2>&1
echo "optimization exit $?"
timeout 200 python -m plenoctree_amd.octree.evaluation --train_dir /tmp/pxo_conv --config /tmp/pxo_conv/cfg.yaml --input /tmp/pxo_conv/tree_opt.npz >> gpurun_out/converge.log 2>&1
echo "evaluation exit $?"
ls -la /tmp/pxo_conv/*.npz >> gpurun_out/converge.log 2>&1
grep -v amdgpu.ids gpurun_out/converge.log | tail -45
