#!/usr/bin/env python
"""Timing of the PlenOctree-side kernels at the reference's sizes (init_grid_depth 8 -> 512^3 grid, 800x800
views) on an analytic scene: three fuzzy spheres' density on the grid, random SH16 coefficients in the leaves.

Prints one JSON line with per-kernel times and, per kernel, a recomputable ROOFLINE ("HBM-bound
pointer-chasing kernel ... different roofline from (a)"): the work counters of pxo_octree_count_work /
pxo_grid_weight_count_work (the kernels' own march, counting) give

  algorithmic bytes = what the kernel must move if nothing were cached: a D-float leaf row per sample above the sigma
                      threshold (196 B for SH16), 4 B of sigma per other sample, 4 B per child pointer of the leaf
                      lookups, 12 B per ray of output (+ for the backward: grad_out and the forward image, 24 B per ray,
                      and every touched gradient row read-modify-written once, 2 x 196 B per DISTINCT leaf)
  hbm floor bytes   = the same with every leaf row fetched ONCE per launch (distinct leaves x 196 B): what a perfect cache
                      would still have to read from HBM
  achieved          = algorithmic bytes / measured time, against PEAK_HBM_ACHIEVABLE = 6.3 TB/s
                      (MI355X_MICROARCH.md: achievable HBM3E rate; nominal 8 TB/s)

Secondary measurement for DESIGN.md -- bench.py's headline stays the training metric.
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


PEAK_HBM_ACHIEVABLE_GBPS = 6300.0        # MI355X_MICROARCH.md: what a streaming kernel reaches
PEAK_HBM_SPEC_GBPS = 8000.0              # the HBM3E specification


SCENE = {}      # what the running measurement works on; filled by measure() as it builds the scene


def scene_mismatch(doc_scene):
    """Which keys of the counters' scene differ from the running one (a tree built from another camera set is another tree:
    round 5 divided the bytes of a 1.06 GB tree by the floor of a 1.4 GB one and reported traffic below the floor)."""
    bad = []
    for k, v in SCENE.items():
        if k not in doc_scene:
            bad.append(f"{k}: not recorded")
        elif isinstance(v, float):
            if abs(doc_scene[k] - v) > 1e-3 * max(abs(v), 1.0):
                bad.append(f"{k}: {doc_scene[k]} vs {v}")
        elif doc_scene[k] != v:
            bad.append(f"{k}: {doc_scene[k]} vs {v}")
    return bad


def hbm_traffic(kernel):
    """Measured HBM bytes per launch of `kernel` from the committed PMC passes of scripts/octree_bench.py
    (profiles/octree_hbm_traffic.json, written by scripts/summarize_octree_prof.py); None if absent -- and a refusal (dict with
    only `refused`) when those passes ran on a different scene than this run's."""
    try:
        with open(os.path.join(ROOT, "profiles", "octree_hbm_traffic.json")) as f:
            doc = json.load(f)
        bad = scene_mismatch(doc.get("scene", {}))
        if bad:
            return {"refused": "profiles/octree_hbm_traffic.json was collected on another scene (" + "; ".join(bad) +
                               "): no traffic ratio is computed from it"}
        k = doc["kernels"][kernel]
        per = float(k.get("cameras_per_launch", 1))          # the weight mask's launch serves several cameras; its roofline is per camera
        return {"read_bytes": k["hbm_read_bytes_per_launch"] / per, "write_bytes": k["hbm_write_bytes_per_launch"] / per,
                "source": "profiles/octree_hbm_traffic.json (" + doc.get("source", "rocprofv3 --pmc passes") + "); not measured inside this run"}
    except Exception:
        return None


def roofline(alg_bytes, floor_bytes, ms, parts, kernel=None):
    gbps = alg_bytes / (ms * 1e-3) / 1e9
    r = {"bound": "hbm", "algorithmic_bytes": alg_bytes, "hbm_floor_bytes": floor_bytes, "ms": ms, "achieved": gbps,
         "peak": PEAK_HBM_ACHIEVABLE_GBPS, "unit": "GB/s", "frac": gbps / PEAK_HBM_ACHIEVABLE_GBPS,
         "frac_of_spec_8TBps": gbps / PEAK_HBM_SPEC_GBPS,
         "frac_of_hbm_floor": floor_bytes / (ms * 1e-3) / 1e9 / PEAK_HBM_ACHIEVABLE_GBPS, "per_image": parts}
    if kernel:
        r["kernel"] = kernel
        t = hbm_traffic(kernel)
        if t and "refused" in t:
            r["traffic"] = None
            r["traffic_refused"] = t["refused"]
        elif t:
            r["traffic"] = t["read_bytes"] + t["write_bytes"]
            r["traffic_over_floor"] = r["traffic"] / max(floor_bytes, 1)
            r["traffic_detail"] = t
    return r


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--depth", type=int, default=8)
    p.add_argument("--size", type=int, default=800)
    p.add_argument("--step", type=float, default=1e-4)
    p.add_argument("--cams", type=int, default=8)
    p.add_argument("--basis", type=int, default=16, help="SH basis_dim of the tree data (16 or 25)")
    p.add_argument("--gw-only", action="store_true", help="stop after grid_weight_render (A/B of that kernel)")
    p.add_argument("--hard", action="store_true", help="exact zeros outside the spheres (as a trained, relu'd density has) instead of fuzzy tails")
    p.add_argument("--no-roofline", action="store_true", help="skip the counting passes")
    p.add_argument("--reps", type=int, default=2)
    p.add_argument("--tune", default="", help="A/B only: pxo_octree_set_tuning knobs, e.g. bwd_update=1,bwd_cache_rows=32,gw_marcher=0")
    a = p.parse_args()
    if a.tune:
        from plenoctree_amd import octree_ops as oops
        knobs = {"gw_marcher": oops.TUNE_GW_MARCHER, "bwd_cache_rows": oops.TUNE_BWD_CACHE_ROWS, "bwd_update": oops.TUNE_BWD_UPDATE,
                 "gw_tile_order": oops.TUNE_GW_TILE_ORDER}
        for kv in a.tune.split(","):
            k, _, v = kv.partition("=")
            oops.set_tuning(knobs[k], int(v))
    out = measure(a)
    if a.tune:
        out["tuning"] = a.tune
    print(json.dumps(out), flush=True)


def defaults(**over):
    """The argument namespace of `measure` for callers that are not this CLI (bench.py's `octree` record)."""
    a = argparse.Namespace(depth=8, size=800, step=1e-4, cams=8, basis=16, gw_only=False, hard=False, no_roofline=False, reps=2, tune="")
    for k, v in over.items():
        setattr(a, k, v)
    return a


def measure(a):
    from plenoctree_amd import build, octree_ops as oops
    from plenoctree_amd.nerf_sh.nerf.datasets import pose_spherical
    from plenoctree_amd.octree.svox import N3Tree, VolumeRenderer
    build.build(verbose=False)
    dev = torch.device("cuda", torch.cuda.current_device())      # the caller's device (bench.py: one rank per GPU)
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
    SCENE.clear()
    SCENE.update(basis_dim=K, reso=reso, image=[H, W], cams=a.cams, step_size=float(a.step), hard=bool(a.hard))

    opts = oops.render_opts(a.step)
    wt = torch.zeros(reso ** 3, device=dev)
    ms = timed(lambda: oops.grid_weight_render(sig, reso, cams, focal, focal, W, H, opts, tree.offset, tree.invradius, grid_weight=wt), reps=1)
    out["grid_weight_render_ms_per_cam"] = ms / a.cams
    out["grid_weight_Mrays_per_s"] = a.cams * W * H / ms / 1e3
    if not a.no_roofline:
        c = oops.grid_weight_count_work(sig, reso, cams, focal, focal, W, H, opts, tree.offset, tree.invradius)
        n = a.cams
        per = {"rays": c["rays"] / n, "samples": c["samples"] / n, "occupied_samples": c["occupied_samples"] / n,
               "distinct_voxels_all_cams": c["distinct_voxels"], "brick_passes_bytes": 4 * reso ** 3 * 4 / n}
        # sigma: 4 B per sample; weights: every touched voxel read-modify-written once per call; bricking sigma and
        # unbricking the weights stream the grid 4 times per call (amortised over the call's cameras)
        alg = per["samples"] * 4 + c["distinct_voxels"] * 8 / n + per["brick_passes_bytes"]
        floor = (min(c["distinct_voxels"] * 4 + c["distinct_voxels"] * 8, reso ** 3 * 12)) / n + per["brick_passes_bytes"]
        out["grid_weight_roofline"] = roofline(alg, floor, ms / a.cams, per, "grid_weight_pow2_kernel")
    mask = oops.threshold_mask(wt, 1e-3)
    out["mask_voxels"] = int(mask.sum())
    out["weight_sum"] = float(wt.double().sum())
    if a.gw_only:
        return out
    t0 = time.perf_counter()
    child, pd, levels = oops.tree_from_mask(mask, depth)
    torch.cuda.synchronize()
    out["tree_build_ms_first"] = (time.perf_counter() - t0) * 1e3
    out["tree_build_ms"] = timed(lambda: oops.tree_from_mask(mask, depth), reps=3)
    tree.refine_from_mask(mask)
    out["n_internal"], out["level_nodes"] = tree.n_internal, tree.level_nodes
    out["tree_data_MB"] = tree.data.numel() * 4 / 1e6
    SCENE.update(n_internal=int(tree.n_internal), tree_data_MB=float(out["tree_data_MB"]))      # the tree depends on the camera set
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
    counts = {}

    def render_all(fast):
        with torch.no_grad():
            return [r.render_persp(c, width=W, height=H, fx=focal, fast=fast) for c in cams]

    for fast in (False, True):
        ms = timed(lambda: render_all(fast), reps=a.reps) / a.cams
        key = "fast" if fast else "exact"
        out[f"render_{key}_ms_per_image"] = ms
        out[f"render_{key}_Mrays_per_s"] = W * H / ms / 1e3
        out[f"render_{key}_fps"] = 1e3 / ms
        if not a.no_roofline:
            cs = [oops.octree_count_work(tree.view(), c, W, H, focal, r._opts(fast)) for c in cams]
            avg = {k: sum(x[k] for x in cs) / len(cs) for k in cs[0]}
            D = tree.data_dim
            alg = avg["shaded_samples"] * D * 4 + (avg["samples"] - avg["shaded_samples"]) * 4 + avg["child_loads"] * 4 + W * H * 12
            floor = avg["distinct_leaves"] * D * 4 + W * H * 12
            out[f"render_{key}_roofline"] = roofline(alg, floor, ms, avg, "octree_render_kernel" if not fast else None)
            counts[key] = (avg, alg, floor)
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
        ms = timed(lambda: bwd(reuse), reps=a.reps) / a.cams
        key = "render_bwd_reusing_fwd" if reuse else "render_bwd"
        out[f"{key}_ms_per_image"] = ms
        out[f"{key}_Mrays_per_s"] = W * H / ms / 1e3
        if "exact" in counts:
            # the backward marches exactly (no early stop): the exact forward's reads per march (two marches without the
            # kept image), + grad_out and the forward image per ray, + every touched gradient row read-modify-written once
            avg, alg_f, floor_f = counts["exact"]
            D = tree.data_dim
            marches = 1 if reuse else 2
            extra = W * H * (24 if reuse else 12) + avg["distinct_leaves"] * D * 4 * 2
            out[f"{key}_roofline"] = roofline(marches * (alg_f - W * H * 12) + extra, floor_f - W * H * 12 + extra, ms,
                                              dict(avg, marches=marches), "octree_render_bwd4_kernel" if reuse else None)
    out["grad_abs_sum"] = float(grad.double().abs().sum())
    out["sgd_ms"] = timed(lambda: oops.sgd_step(tree.data, grad, 0.0), reps=3)
    # SGD streams data + gradient in and data out: 12 B per float
    out["sgd_roofline"] = roofline(tree.data.numel() * 12.0, tree.data.numel() * 12.0, out["sgd_ms"], {}, "sgd_kernel")
    out["scene"] = dict(SCENE)
    return out


if __name__ == "__main__":
    main()
