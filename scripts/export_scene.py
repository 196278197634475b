#!/usr/bin/env python
"""Writes the analytic scene of datasets.Synthetic in the reference's two ON-DISK formats, so that the drop-in CLIs can be run
through the real loaders at full size (no NeRF-Synthetic / Tanks&Temples scene exists on the boxes):

  blender   NeRF-Synthetic directory (nerf_sh/nerf/datasets.py:189-232): transforms_{train,val,test}.json with camera_angle_x and
            per-frame transform_matrix, <split>/r_<i>.png as 8-bit RGBA (alpha = 1 on the spheres, 0 on the background: the
            loader composites on white)
  nsvf      NSVF directory (nerf_sh/nerf/datasets.py:491-552; bbox.txt as octree/nerf/datasets.py:73-77): intrinsics.txt, bbox.txt,
            pose/<p>_<i>.txt (camera-to-world in the OpenCV axes the loader converts FROM), rgb/<p>_<i>.png (8-bit RGB on white),
            prefix 0_ train / 1_ val / 2_ test

The pixel values are those of datasets.Synthetic with synthetic_8bit (colours rounded to k / 255 -- what a PNG holds) and the
poses its float32 matrices printed with 9 significant digits (exact round trip), so a run on these files and a run on the
Synthetic class with the same options see the SAME bits: rays, pixels, batches (tests/test_gpu_formats.py).  Runs on the GPU
box (the scene is evaluated by the device feeder).

  python scripts/export_scene.py blender /tmp/scene_blender --size 800 800 --views 100 8 8
  python scripts/export_scene.py nsvf /tmp/scene_nsvf --size 1080 1920 --views 100 8 8
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def synthetic_args(h, w, n_train, n_test):
    from plenoctree_amd.nerf_sh.nerf import utils
    args = utils.define_flags().parse_args(["--config", "synthetic", "--train_dir", "x"])
    utils.update_flags(args)
    args.factor = 0
    args.synthetic_hw = (h, w)
    args.synthetic_views = (n_train, n_test)
    args.synthetic_8bit = True
    return args


def views(split, args, device, count):
    """[(c2w float32 [4,4], rgb uint8 [H,W,3], alpha uint8 [H,W])] of the first `count` views of Synthetic's `split` poses."""
    from plenoctree_amd.nerf_sh.nerf import datasets
    a = argparse.Namespace(**vars(args))
    if split == "train":          # the training views are read one by one below: no need for the resident copy
        ds = datasets.Synthetic("test", a, device)         # same class; poses re-seeded as the train split's
        rs = np.random.RandomState(7)
        ds.camtoworlds = np.stack([datasets.pose_spherical(rs.uniform(0, 360), rs.uniform(-10, 60), 4.0311) for _ in range(args.synthetic_views[0])])
        ds._c2w_dev = torch.from_numpy(np.ascontiguousarray(ds.camtoworlds[:, :3, :4])).to(device)
        ds.n_examples = len(ds.camtoworlds)
    else:
        ds = datasets.Synthetic("test", a, device)
    ids = torch.arange(ds.h * ds.w, device=device)
    out = []
    for i in range(count):
        rays = ds._rays_for(i, ids)
        rgb, alpha = datasets.analytic_scene_rgb(rays.origins, rays.directions, True, return_alpha=True)
        rgb8 = torch.round(rgb * 255.0).to(torch.uint8).reshape(ds.h, ds.w, 3).cpu().numpy()
        a8 = (alpha * 255.0).to(torch.uint8).reshape(ds.h, ds.w).cpu().numpy()
        out.append((ds.camtoworlds[i].astype(np.float32), rgb8, a8))
    return out, ds


def write_blender(path, h, w, n_train, n_val, n_test, device):
    from PIL import Image
    args = synthetic_args(h, w, n_train, max(n_test, n_val))
    os.makedirs(path, exist_ok=True)
    for split, count in (("train", n_train), ("val", n_val), ("test", n_test)):
        vs, ds = views("train" if split == "train" else "test", args, device, count)
        os.makedirs(os.path.join(path, split), exist_ok=True)
        frames = []
        for i, (c2w, rgb8, a8) in enumerate(vs):
            rgba = np.concatenate([np.where(a8[..., None] > 0, rgb8, 0).astype(np.uint8), a8[..., None]], -1)
            Image.fromarray(rgba, "RGBA").save(os.path.join(path, split, f"r_{i}.png"), compress_level=1)
            frames.append({"file_path": f"./{split}/r_{i}", "rotation": 0.0,
                           "transform_matrix": [[float(np.float32(x)) for x in row] for row in c2w]})
        with open(os.path.join(path, f"transforms_{split}.json"), "w") as f:
            json.dump({"camera_angle_x": 0.6911112, "frames": frames}, f, indent=1)
    return {"format": "blender", "path": path, "size": [h, w], "views": [n_train, n_val, n_test]}


def write_nsvf(path, h, w, n_train, n_val, n_test, device):
    from PIL import Image
    args = synthetic_args(h, w, n_train, max(n_test, n_val))
    for sub in ("pose", "rgb"):
        os.makedirs(os.path.join(path, sub), exist_ok=True)
    focal = None
    cam_trans = np.diag(np.array([1, -1, -1, 1], dtype=np.float64))       # the loader multiplies by this: it is its own inverse
    for prefix, split, count in (("0_", "train", n_train), ("1_", "val", n_val), ("2_", "test", n_test)):
        vs, ds = views("train" if split == "train" else "test", args, device, count)
        focal = ds.focal
        for i, (c2w, rgb8, a8) in enumerate(vs):
            Image.fromarray(rgb8, "RGB").save(os.path.join(path, "rgb", f"{prefix}{i:04d}.png"), compress_level=1)
            np.savetxt(os.path.join(path, "pose", f"{prefix}{i:04d}.txt"), c2w.astype(np.float64) @ cam_trans, fmt="%.9g")
    K = np.eye(4)
    K[0, 0] = K[1, 1] = focal
    K[0, 2], K[1, 2] = 0.5 * w, 0.5 * h
    np.savetxt(os.path.join(path, "intrinsics.txt"), K, fmt="%.17g")
    with open(os.path.join(path, "bbox.txt"), "w") as f:                  # xmin ymin zmin xmax ymax zmax voxel_size
        f.write("-1.2 -1.2 -1.2 1.2 1.2 1.2 0.1\n")
    return {"format": "nsvf", "path": path, "size": [h, w], "views": [n_train, n_val, n_test], "focal": focal}


def main():
    p = argparse.ArgumentParser()
    p.add_argument("format", choices=["blender", "nsvf"])
    p.add_argument("path")
    p.add_argument("--size", type=int, nargs=2, default=None, metavar=("H", "W"))
    p.add_argument("--views", type=int, nargs=3, default=[100, 8, 8], metavar=("TRAIN", "VAL", "TEST"))
    a = p.parse_args()
    dev = torch.device("cuda:0")
    h, w = a.size or ((800, 800) if a.format == "blender" else (1080, 1920))
    fn = write_blender if a.format == "blender" else write_nsvf
    print(json.dumps(fn(a.path, h, w, *a.views, dev)))


if __name__ == "__main__":
    main()
