#!/usr/bin/env python
"""A/B of the fused MLP kernels in float32 and in bf16x6 on one GPU: HIP-event time per launch of mlp_fwd (saved tensors) and
mlp_bwd_data at the BASELINE batch (4096 rays x 64 + 192 samples + 10k sparsity points), and the whole train step, dense and
with zero-row skipping.  Random Glorot weights (the kernels' time does not depend on the values, except in skipping mode).

  python scripts/x6_probe.py [--batch 4096] [--steps 10] [--sigma-shift -2.0]
One JSON line per variant on stdout."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--sigma-shift", type=float, default=0.0)
    ap.add_argument("--variants", type=str, default="f32,x6,f32_skip,x6_skip")
    ap.add_argument("--x6-wgrad", type=int, default=None, help="pxo_set_tuning(PXO_TUNE_X6_WGRAD, n) before the runs (A/B)")
    args = ap.parse_args()
    from plenoctree_amd import ops
    from oracle import nerf_oracle as O           # parameter initialisation of the test helpers only
    from _helpers import make_params, make_rays, pxo_cfg, split_mlp
    dev = torch.device("cuda:0")
    if args.x6_wgrad is not None:
        ops.set_tuning(ops.TUNE_X6_WGRAD, args.x6_wgrad)
    cfg = O.Cfg()
    flat = make_params(cfg, bias_scale=0.2)
    n = flat.numel() // 2
    b8 = sum(fi * fo + fo for fi, fo in O.layer_shapes(cfg)[:8]) + O.layer_shapes(cfg)[8][0]
    for mi in range(2):
        flat[mi * n + b8] += args.sigma_shift
    fd = flat.to(dev)
    B = args.batch
    rays = make_rays(B, 5)
    o, d, v = rays.origins.to(dev), rays.directions.to(dev), rays.viewdirs.to(dev)
    px = torch.rand(B, 3, device=dev)
    M = {"coarse": B * 64, "fine": B * 192 + cfg.sparsity_npoints}
    for name in args.variants.split(","):
        pcfg = pxo_cfg(ops, cfg)
        pcfg.mlp_precision = 2 if name.startswith("x6") else 0
        pcfg.skip_zero_rows = 1 if name.endswith("skip") else 0
        packed = [ops.pack_weights(pcfg, split_mlp(fd, cfg, i)) for i in range(2)]
        grads = torch.zeros_like(fd); stats = torch.zeros(6, device=dev)
        ws = torch.empty(ops.train_workspace_bytes(pcfg, B), dtype=torch.uint8, device=dev)

        def step(seed):
            ops.train_fwd_bwd(pcfg, fd, packed, o, d, v, px, grads, stats, ws, randomized=True, seed=seed)
        for i in range(3):
            step(i)
        torch.cuda.synchronize()
        t0 = time.time()
        for i in range(args.steps):
            step(10 + i)
        torch.cuda.synchronize()
        ms_step = (time.time() - t0) * 1e3 / args.steps
        ops.profile_enable(True)
        for i in range(args.steps):
            step(100 + i)
        torch.cuda.synchronize()
        rec = {"variant": name, "batch": B, "ms_per_step": round(ms_step, 4), "rays_per_s": round(B / ms_step * 1e3)}
        for tag, key in ((ops.PROF_MLP_FWD, "mlp_fwd"), (ops.PROF_MLP_BWD_DATA, "mlp_bwd_data"), (ops.PROF_WGRAD_MAIN, "wgrad_main"),
                         (ops.PROF_WGRAD_OTHER, "wgrad_other")):
            nl, ms, rows = ops.profile_read(tag)
            rec[key + "_ms_per_launch"] = round(ms / max(nl, 1), 4)
            rec[key + "_launches_per_step"] = nl / args.steps
        ops.profile_enable(False)
        live, total = ops.train_backward_work(pcfg, B, ws)
        rec["live_chunk_fraction"] = round(live / total, 4)
        rec["loss"] = float(stats[0])
        print(json.dumps(rec), flush=True)
        del ws


if __name__ == "__main__":
    main()
