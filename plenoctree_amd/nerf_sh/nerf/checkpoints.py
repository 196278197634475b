"""Checkpoint save/restore for TrainState (the role of flax.training.checkpoints in
nerf_sh/train.py:240-242 and nerf_sh/nerf/models.py:46-48).  Files are `checkpoint_<step>`
torch archives holding the flat arena, the Adam moments and the step; `keep` bounds how
many are retained (reference: keep=200)."""
import glob
import os
import re

import torch


def _step_of(path):
    m = re.search(r"checkpoint_(\d+)$", path)
    return int(m.group(1)) if m else -1


def latest_checkpoint(train_dir):
    paths = sorted(glob.glob(os.path.join(train_dir, "checkpoint_*")), key=_step_of)
    paths = [p for p in paths if _step_of(p) >= 0]
    return paths[-1] if paths else None


def save_checkpoint(train_dir, state, step, keep=200):
    os.makedirs(train_dir, exist_ok=True)
    path = os.path.join(train_dir, f"checkpoint_{int(step)}")
    tmp = path + ".tmp"
    torch.save(state.state_dict(), tmp)
    os.replace(tmp, path)
    paths = sorted((p for p in glob.glob(os.path.join(train_dir, "checkpoint_*")) if _step_of(p) >= 0), key=_step_of)
    for old in paths[:-keep]:
        os.remove(old)
    return path


def restore_checkpoint(train_dir, state):
    """Loads the newest checkpoint into `state` in place; returns the path or None."""
    path = latest_checkpoint(train_dir)
    if path is None:
        return None
    state.load_state_dict(torch.load(path, map_location="cpu"))
    return path
