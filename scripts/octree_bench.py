#!/usr/bin/env python
"""Timing of the PlenOctree-side kernels at the reference's sizes (init_grid_depth 8 -> 512^3 grid, 800x800
views) on an analytic scene: three fuzzy spheres' density on the grid, random SH16 coefficients in the leaves.

Prints one JSON line with per-kernel times and achieved algorithmic rates (rays/s, leaf visits/s, GB/s of leaf
data touched).  Secondary measurement for DESIGN.md -- bench.py's headline stays the training metric.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--depth", type=int, default=8)
    p.add_argument("--size", type=int, default=800)
    p.add_argument("--step", type=float, default=1e-4)
    p.add_argument("--cams", type=int, default=8)
    p.add_argument("--basis", type=int, default=16, help="SH basis_dim of the tree data (16 or 25)")
    p.add_argument("--gw-only", action="store_true", help="stop after grid_weight_render (A/B of that kernel)")
    p.add_argument("--hard", action="store_true", help="exact zeros outside the spheres (as a trained, relu'd density has) instead of fuzzy tails")
    a = p.parse_args()
    from plenoctree_amd import build, octree_ops as oops
    from plenoctree_amd.nerf_sh.nerf.datasets import pose_spherical
    from plenoctree_amd.octree.svox import N3Tree, VolumeRenderer
    build.build(verbose=False)
    dev = torch.device("cuda:0")
    depth, reso = a.depth, 2 ** (a.depth + 1)
    K = a.basis
    tree = N3Tree(N=2, data_dim=3 * K + 1, depth_limit=depth, radius=1.5, center=[0, 0, 0], data_format=f"SH{K}", map_location=dev)
    # density of three fuzzy spheres on the grid (world coords in [-1.5, 1.5]^3)
    ax = ((torch.arange(reso, device=dev, dtype=torch.float32) + 0.5) / reso - 0.5) * 3.0
    sig = torch.zeros(reso, reso, reso, device=dev)
    for c, r in (((0.0, 0.0, 0.0), 0.6), ((0.7, 0.3, 0.2), 0.3), ((-0.5, -0.4, 0.5), 0.35)):
        d2 = (ax[:, None, None] - c[0]) ** 2 + (ax[None, :, None] - c[1]) ** 2 + (ax[None, None, :] - c[2]) ** 2
        sig += 40.0 * torch.sigmoid((r - d2.sqrt()) * 60.0)
    if a.hard:
        sig = torch.where(sig > 1.0, sig, torch.zeros_like(sig))
    sig = sig.reshape(-1).contiguous()
    out_occ = float((sig > 0).float().mean())
    W = H = a.size
    focal = 0.5 * W / np.tan(0.5 * 0.6911112)
    rs = np.random.RandomState(7)
    cams = torch.from_numpy(np.stack([pose_spherical(rs.uniform(0, 360), rs.uniform(-10, 60), 4.0311) for _ in range(a.cams)])).to(dev)
    out = {"sigma_positive_fraction": out_occ, "basis_dim": K, "depth": depth, "reso": reso, "image": [H, W], "step_size": a.step, "cams": a.cams}

    opts = oops.render_opts(a.step)
    wt = torch.zeros(reso ** 3, device=dev)
    ms = timed(lambda: oops.grid_weight_render(sig, reso, cams, focal, focal, W, H, opts, tree.offset, tree.invradius, grid_weight=wt), reps=1)
    out["grid_weight_render_ms_per_cam"] = ms / a.cams
    out["grid_weight_Mrays_per_s"] = a.cams * W * H / ms / 1e3
    mask = oops.threshold_mask(wt, 1e-3)
    out["mask_voxels"] = int(mask.sum())
    out["weight_sum"] = float(wt.double().sum())
    if a.gw_only:
        print(json.dumps(out), flush=True)
        return
    t0 = time.perf_counter()
    child, pd, levels = oops.tree_from_mask(mask, depth)
    torch.cuda.synchronize()
    out["tree_build_ms_first"] = (time.perf_counter() - t0) * 1e3
    out["tree_build_ms"] = timed(lambda: oops.tree_from_mask(mask, depth), reps=3)
    tree.refine_from_mask(mask)
    out["n_internal"], out["level_nodes"] = tree.n_internal, tree.level_nodes
    node0, count = tree.max_depth_nodes()
    out["sample_cells_ms"] = timed(lambda: tree.sample_max_depth_cells(8, seed=1), reps=3)
    out["sample_points"] = count * 8 * 8
    # leaf data: sigma from the grid at the cell centre, random SH
    pts = tree.sample_max_depth_cells(1, u=torch.full((count * 8, 3), 0.5, device=dev)).view(-1, 3)
    idx = ((pts / 3.0 + 0.5) * reso).long().clamp_(0, reso - 1)
    leaf = tree.max_depth_data()
    leaf.copy_(torch.randn(leaf.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(0)) * 0.5)
    leaf[:, -1] = sig[(idx[:, 0] * reso + idx[:, 1]) * reso + idx[:, 2]]
    r = VolumeRenderer(tree, step_size=a.step)
    def render_all(fast):
        with torch.no_grad():
            return [r.render_persp(c, width=W, height=H, fx=focal, fast=fast) for c in cams]

    for fast in (False, True):
        ms = timed(lambda: render_all(fast), reps=2) / a.cams
        key = "fast" if fast else "exact"
        out[f"render_{key}_ms_per_image"] = ms
        out[f"render_{key}_Mrays_per_s"] = W * H / ms / 1e3
        out[f"render_{key}_fps"] = 1e3 / ms
    with torch.no_grad():
        im = r.render_persp(cams[0], width=W, height=H, fx=focal)
    out["image_mean"] = float(im.mean())
    out["image_sum"] = float(im.double().sum())
    gt = torch.rand_like(im)
    grad = torch.zeros_like(tree.data)

    ims = render_all(False)

    def bwd(reuse):
        for c, imc in zip(cams, ims):
            _, g = oops.image_mse(imc, gt)
            oops.octree_render_persp_bwd(tree.view(), c, W, H, focal, r._opts(False), g, grad, out_rgb=imc if reuse else None)
    for reuse in (False, True):
        ms = timed(lambda: bwd(reuse), reps=2) / a.cams
        key = "render_bwd_reusing_fwd" if reuse else "render_bwd"
        out[f"{key}_ms_per_image"] = ms
        out[f"{key}_Mrays_per_s"] = W * H / ms / 1e3
    out["grad_abs_sum"] = float(grad.double().abs().sum())
    out["sgd_ms"] = timed(lambda: oops.sgd_step(tree.data, grad, 0.0), reps=3)
    out["tree_data_MB"] = tree.data.numel() * 4 / 1e6
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
