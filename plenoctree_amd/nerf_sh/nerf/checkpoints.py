"""Checkpoint save/restore in the flax msgpack format of the reference.

The reference saves `flax.training.checkpoints.save_checkpoint(train_dir, state, step, keep=200)`
(nerf_sh/train.py:240-242, 306-310) and reads the files back in three places: resume
(nerf_sh/nerf/models.py:46-48), eval (nerf_sh/eval.py:70) and the PlenOctree extraction, whose
consumer fixes the key names: ckpt["optimizer"]["target"]["params"]["MLP_i"]["Dense_j"]{"kernel",
"bias"} with kernels [in,out] (octree/nerf/models.py:75-102).

flax (>=0.3.1, environment.yml:19) is a third-party dependency that is not installed here; its
published serialisation (flax/serialization.py) is restated: the file `checkpoint_<step>` is
msgpack of the nested state dict, every ndarray leaf packed as ExtType(1, msgpack((shape, dtype
name, C-order bytes))).  State dict of TrainState(optimizer=flax.optim.Optimizer):
  {"optimizer": {"target": {"params": ...},
                 "state": {"step": int32 scalar,
                           "param_states": {"params": <same tree>{"grad_ema", "grad_sq_ema"}}}}}
Unverified against a live flax (none available); tests/test_checkpoint_cpu.py drives the
reference's own consumer code with these files instead.
"""
import glob
import os
import re

import msgpack
import numpy as np
import torch

_EXT_NDARRAY = 1
_EXT_NPSCALAR = 3


def _pack_ndarray(arr):
    arr = np.ascontiguousarray(arr)
    return msgpack.packb((list(arr.shape), arr.dtype.name, arr.tobytes("C")), use_bin_type=True)


def _ext_pack(x):
    if isinstance(x, np.ndarray):
        return msgpack.ExtType(_EXT_NDARRAY, _pack_ndarray(x))
    if isinstance(x, np.generic):
        return msgpack.ExtType(_EXT_NPSCALAR, _pack_ndarray(np.asarray(x)))
    raise TypeError(f"cannot serialise {type(x)}")


def _ext_unpack(code, data):
    if code in (_EXT_NDARRAY, _EXT_NPSCALAR):
        shape, dtype_name, buf = msgpack.unpackb(data, raw=False)
        arr = np.frombuffer(buf, dtype=np.dtype(dtype_name)).reshape(shape).copy()
        return arr[()] if code == _EXT_NPSCALAR else arr
    return msgpack.ExtType(code, data)


def msgpack_serialize(tree):
    return msgpack.packb(tree, default=_ext_pack, strict_types=True)


def msgpack_restore(blob):
    return msgpack.unpackb(blob, ext_hook=_ext_unpack, raw=False)


def _leaves(cfg):
    from ... import ops
    return ops.param_layout(cfg)


def arena_to_tree(flat, cfg):
    """flat 2-MLP arena (numpy) -> {"MLP_0": {"Dense_0": {"kernel": [in,out], "bias": [out]}, ...}, "MLP_1": ...}"""
    leaves, n = _leaves(cfg)
    tree = {}
    for mi in range(2):
        mlp = {}
        for layer, is_bias, off, rows, cols in leaves:
            d = mlp.setdefault(f"Dense_{layer}", {})
            a = flat[mi * n + off: mi * n + off + rows * cols]
            d["bias" if is_bias else "kernel"] = a.copy() if is_bias else a.reshape(rows, cols).copy()
        tree[f"MLP_{mi}"] = mlp
    return tree


def tree_to_arena(tree, cfg, leaf=None):
    """Inverse of arena_to_tree; `leaf` picks a sub-key (e.g. "grad_ema") of every array leaf."""
    leaves, n = _leaves(cfg)
    flat = np.zeros(2 * n, np.float32)
    for mi in range(2):
        for layer, is_bias, off, rows, cols in leaves:
            a = tree[f"MLP_{mi}"][f"Dense_{layer}"]["bias" if is_bias else "kernel"]
            if leaf is not None:
                a = a[leaf]
            a = np.asarray(a, np.float32)
            want = (rows,) if is_bias else (rows, cols)
            if tuple(a.shape) != want:
                raise ValueError(f"checkpoint leaf MLP_{mi}/Dense_{layer} has shape {a.shape}, expected {want}")
            flat[mi * n + off: mi * n + off + rows * cols] = a.reshape(-1)
    return flat


def state_to_tree(state):
    p = state.params.detach().cpu().numpy()
    m = state.m.detach().cpu().numpy()
    v = state.v.detach().cpu().numpy()
    params = arena_to_tree(p, state.cfg)
    mt, vt = arena_to_tree(m, state.cfg), arena_to_tree(v, state.cfg)
    pstates = {mk: {dk: {lk: {"grad_ema": mt[mk][dk][lk], "grad_sq_ema": vt[mk][dk][lk]} for lk in dv}
                    for dk, dv in mv.items()} for mk, mv in params.items()}
    return {"optimizer": {"target": {"params": params},
                          "state": {"step": np.asarray(state.step, np.int32), "param_states": {"params": pstates}}}}


def load_tree_into_state(tree, state):
    opt = tree["optimizer"]
    params = opt["target"]["params"]
    dev = state.params.device
    state.params.copy_(torch.from_numpy(tree_to_arena(params, state.cfg)).to(dev))
    ps = opt.get("state", {}).get("param_states", {}).get("params")
    if ps is not None:
        state.m.copy_(torch.from_numpy(tree_to_arena(ps, state.cfg, "grad_ema")).to(dev))
        state.v.copy_(torch.from_numpy(tree_to_arena(ps, state.cfg, "grad_sq_ema")).to(dev))
    state.step = int(np.asarray(opt.get("state", {}).get("step", 0)).reshape(-1)[0])
    state.repack()


def _step_of(path):
    m = re.search(r"checkpoint_(\d+)$", path)
    return int(m.group(1)) if m else -1


def latest_checkpoint(train_dir):
    paths = [p for p in glob.glob(os.path.join(train_dir, "checkpoint_*")) if _step_of(p) >= 0]
    return max(paths, key=_step_of) if paths else None


def save_checkpoint(train_dir, state, step, keep=200):
    """flax.training.checkpoints.save_checkpoint(train_dir, state, step, keep)."""
    os.makedirs(train_dir, exist_ok=True)
    path = os.path.join(train_dir, f"checkpoint_{int(step)}")
    tmp = path + ".tmp"
    with open(tmp, "wb") as f:
        f.write(msgpack_serialize(state_to_tree(state)))
    os.replace(tmp, path)
    paths = sorted((p for p in glob.glob(os.path.join(train_dir, "checkpoint_*")) if _step_of(p) >= 0), key=_step_of)
    for old in paths[:-keep]:
        os.remove(old)
    return path


def restore_checkpoint(train_dir, state=None):
    """flax.training.checkpoints.restore_checkpoint: newest `checkpoint_<step>` in train_dir (or the
    file itself).  With `state` the arrays are loaded in place and the path is returned (None if
    there is no checkpoint); with state=None the raw state dict is returned (flax's target=None)."""
    path = train_dir if os.path.isfile(train_dir) else latest_checkpoint(train_dir)
    if path is None:
        return None
    with open(path, "rb") as f:
        tree = msgpack_restore(f.read())
    if state is None:
        return tree
    load_tree_into_state(tree, state)
    return path


# ---- the reference's OTHER checkpoint format: a torch state dict of its torch twin ------------------------------------------
def torch_state_dict_to_tree(sd, depth=8):
    """`ckpt["model"]` of octree/nerf/models.py:52-63 (restore_model_state: `*.ckpt` files holding the state dict of the torch
    NerfModel, whose Linear weights are [out, in]) as the flax params tree: the inverse of the key map the reference applies to
    a flax checkpoint (octree/nerf/models.py:79-102: Dense_i -> input_layers.i for i < net_depth, then sigma_layer, then --
    without view directions -- rgb_layer; kernel = weight.T)."""
    names = [f"input_layers.{i}" for i in range(depth)] + ["sigma_layer", "rgb_layer"]
    for k in sd:
        if any(t in k for t in ("bottleneck_layer", "condition_layers", "sg_lambda", "sg_mu_spher")):
            raise ValueError(f"torch checkpoint key {k!r}: the view-conditioned head / SG basis is not built on the MI355X path "
                             "(use_viewdirs=false SH models only)")
    tree = {}
    for mi in range(2):
        mlp = {}
        for li, name in enumerate(names):
            wk, bk = f"MLP_{mi}.{name}.weight", f"MLP_{mi}.{name}.bias"
            if wk not in sd or bk not in sd:
                raise ValueError(f"torch checkpoint has no {wk} / {bk}")
            w = sd[wk].detach().cpu().numpy() if hasattr(sd[wk], "detach") else np.asarray(sd[wk])
            b = sd[bk].detach().cpu().numpy() if hasattr(sd[bk], "detach") else np.asarray(sd[bk])
            mlp[f"Dense_{li}"] = {"kernel": np.ascontiguousarray(w.T.astype(np.float32)), "bias": b.astype(np.float32)}
        tree[f"MLP_{mi}"] = mlp
    return tree


def latest_torch_checkpoint(train_dir):
    paths = sorted(glob.glob(os.path.join(train_dir, "*.ckpt")))       # octree/nerf/models.py:56-59: sorted, last one
    return paths[-1] if paths else None


def restore_torch_checkpoint(train_dir, state, trust_pickle=False):
    """restore_model_state (octree/nerf/models.py:52-63): the newest `*.ckpt` of train_dir into `state` (parameters only; a
    state dict carries no optimizer moments).  Returns the path, or None when there is no such file.  The file is read with
    weights_only=True; one that also holds non-tensor objects needs trust_pickle=True (the reference's plain torch.load)."""
    path = train_dir if os.path.isfile(train_dir) else latest_torch_checkpoint(train_dir)
    if path is None:
        return None
    try:
        ckpt = torch.load(path, map_location="cpu", weights_only=True)
    except Exception as e:            # pickle.UnpicklingError and friends: non-tensor extras (argparse.Namespace, numpy scalars)
        if not trust_pickle:
            raise ValueError(
                f"{path}: holds more than tensors ({type(e).__name__}: {str(e).splitlines()[0][:200]}).  The reference loads its "
                "*.ckpt files with a plain torch.load, i.e. it unpickles arbitrary objects; this loader does that only when "
                "asked to: restore_torch_checkpoint(..., trust_pickle=True) / --trust_ckpt_pickle true, for files you trust") from e
        ckpt = torch.load(path, map_location="cpu", weights_only=False)
    if not isinstance(ckpt, dict) or "model" not in ckpt:
        raise ValueError(f'{path}: not a checkpoint of the reference\'s torch twin (no "model" state dict)')
    from ... import _lib
    params = torch_state_dict_to_tree(ckpt["model"], depth=_lib.NET_DEPTH)      # the depth the kernels (and state.cfg's arena) are built for
    state.params.copy_(torch.from_numpy(tree_to_arena(params, state.cfg)).to(state.params.device))
    state.m.zero_(); state.v.zero_()
    state.repack()
    return path
