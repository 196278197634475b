#!/usr/bin/env python
"""Weight-mask stage of bench.py's grid512 record (sigma grid of the fixed-seed NeRF after 105 steps, 100 training views
in one pxo_grid_weight_render call) under several settings of the kernel's run-time switches.
usage: gw_scene_bench.py "marcher=0" "marcher=1" "" ...   (one timing per argument; "" = chosen on the device;
marcher -> octree_ops.set_tuning(TUNE_GW_MARCHER, v), tile_order -> TUNE_GW_TILE_ORDER)"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    settings = sys.argv[1:] or [""]
    a = bench.parse(["--steps", "2", "--warmup", "1", "--no-cpu-baseline"])
    job = bench.Job(a)
    tr = bench.run_train(job, a.preset, a.steps, a.warmup, snapshot_step=a.eval_step)
    from plenoctree_amd import octree_ops as oops
    from plenoctree_amd.octree import extraction
    from plenoctree_amd.octree.svox import N3Tree
    model, state, dataset = tr["model"], tr["eval_state"], tr["dataset"]
    comm = job.comm()
    reso, center, radius = 512, [0.0, 0.0, 0.0], [1.5, 1.5, 1.5]
    state.repack(need_bwd=False)
    sig = extraction.grid_sigma(model, state, reso, center, radius, comm)
    tree = N3Tree(N=2, data_dim=49, init_refine=0, depth_limit=8, radius=radius, center=center, data_format="SH16",
                  map_location=job.device)
    out = [{"sigma_positive_fraction": float((sig > 0).float().mean()), "sigma_gt_1": float((sig > 1).float().mean())}]
    default_order = oops.get_tuning(oops.TUNE_GW_TILE_ORDER)
    for rep in range(2):
        for st in settings:
            for kv in st.split(","):
                if kv:
                    k, v = kv.split("=")
                    oops.set_tuning({"marcher": oops.TUNE_GW_MARCHER, "tile_order": oops.TUNE_GW_TILE_ORDER}[k], int(v))
            job.sync()
            t0 = time.perf_counter()
            w = extraction.calculate_grid_weights(dataset, sig, reso, tree.invradius, tree.offset, 1e-4, comm)
            job.sync()
            dt = time.perf_counter() - t0
            out.append({"setting": st, "rep": rep, "ms": 1e3 * dt, "voxels": int((w >= 1e-3).sum()), "sum": float(w.double().sum())})
            oops.set_tuning(oops.TUNE_GW_MARCHER, -1); oops.set_tuning(oops.TUNE_GW_TILE_ORDER, default_order)
            del w
    for o in out:
        print(json.dumps(o), flush=True)


if __name__ == "__main__":
    main()
