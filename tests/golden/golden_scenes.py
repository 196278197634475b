"""Tiny deterministic on-disk scenes (Blender transforms_*.json layout and NSVF layout) shared by
make_golden.py (which runs the REFERENCE's loaders on them) and the tests (which run ours on identical files)."""
import json
import os

import numpy as np
from PIL import Image

H, W = 6, 8


def poses(n, seed):
    rs = np.random.RandomState(seed)
    out = []
    for _ in range(n):
        m = np.eye(4)
        m[:3, :3] = np.linalg.qr(rs.randn(3, 3))[0]
        m[:3, 3] = rs.randn(3) * 2.0
        out.append(m)
    return np.stack(out).astype(np.float32)


def _rgba(rs):
    return (rs.rand(H, W, 4) * 255).astype(np.uint8)


def write_blender(root):
    os.makedirs(root, exist_ok=True)
    rs = np.random.RandomState(11)
    for split, n in (("train", 3), ("val", 1), ("test", 2)):
        frames = []
        os.makedirs(os.path.join(root, split), exist_ok=True)
        for i, pose in enumerate(poses(n, seed=len(split))):
            Image.fromarray(_rgba(rs), "RGBA").save(os.path.join(root, split, f"r_{i}.png"))
            frames.append({"file_path": f"./{split}/r_{i}", "transform_matrix": pose.astype(float).tolist()})
        with open(os.path.join(root, f"transforms_{split}.json"), "w") as f:
            json.dump({"camera_angle_x": 0.6911112070083618, "frames": frames}, f)
    return root


def write_nsvf(root):
    os.makedirs(os.path.join(root, "pose"), exist_ok=True)
    os.makedirs(os.path.join(root, "rgb"), exist_ok=True)
    rs = np.random.RandomState(12)
    K = np.eye(4)
    K[0, 0], K[1, 1], K[0, 2], K[1, 2] = 30.0, 32.0, W / 2, H / 2
    np.savetxt(os.path.join(root, "intrinsics.txt"), K)
    np.savetxt(os.path.join(root, "bbox.txt"), np.array([[-1.0, -2.0, -3.0, 1.0, 2.0, 3.0, 0.4]]))
    names = ["0_0000", "0_0001", "0_0002", "1_0000", "2_0000", "2_0001"]
    for name, pose in zip(names, poses(len(names), seed=3)):
        np.savetxt(os.path.join(root, "pose", f"{name}.txt"), pose)
        Image.fromarray(_rgba(rs), "RGBA").save(os.path.join(root, "rgb", f"{name}.png"))
    return root
