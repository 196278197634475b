"""ctypes binding of libplenoctree_hip.so (include/plenoctree_hip.h, include/plenoctree_octree.h).

The library is the product path: there is no CPU fallback.  Loading fails loudly if the
shared object is missing, and every call raises PxoError on a non-zero status.
"""
import ctypes
import os
from ctypes import (POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_int64, c_size_t,
                    c_uint64, c_void_p)

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libplenoctree_hip.so")

NET_DEPTH = 8
NET_WIDTH = 256
ENC_DIM = 63
ENC_PAD = 64
NUM_LEAVES = 20
MLP_F32, MLP_BF16X3, MLP_BF16X6 = 0, 1, 2


class PxoError(RuntimeError):
    pass


class PxoCfg(Structure):
    _fields_ = [
        ("num_coarse_samples", c_int32),
        ("num_fine_samples", c_int32),
        ("sh_deg", c_int32),
        ("min_deg_point", c_int32),
        ("max_deg_point", c_int32),
        ("white_bkgd", c_int32),
        ("lindisp", c_int32),
        ("sparsity_npoints", c_int32),
        ("near_", c_float),
        ("far_", c_float),
        ("sparsity_weight", c_float),
        ("sparsity_length", c_float),
        ("sparsity_radius", c_float),
        ("weight_decay_mult", c_float),
        ("mlp_precision", c_int32),
        ("noise_std", c_float),
        ("skip_zero_rows", c_int32),
    ]


class PxoLeaf(Structure):
    _fields_ = [
        ("layer", c_int32),
        ("is_bias", c_int32),
        ("offset", c_int64),
        ("rows", c_int32),
        ("cols", c_int32),
    ]


class PxoRenderOpts(Structure):
    _fields_ = [("step_size", c_float), ("background_brightness", c_float), ("sigma_thresh", c_float),
                ("stop_thresh", c_float)]


class PxoTree(Structure):
    _fields_ = [("child", c_void_p), ("data", c_void_p), ("n_internal", c_int64), ("data_dim", c_int32),
                ("basis_dim", c_int32), ("offset", c_float * 3), ("invradius", c_float * 3)]


class PxoCamera(Structure):
    _fields_ = [("c2w", c_void_p), ("fx", c_float), ("fy", c_float), ("width", c_int32), ("height", c_int32)]


TREE_MAX_DEPTH = 10
ABI_VERSION = 6                     # PXO_ABI_VERSION of include/plenoctree_hip.h

P = c_void_p
CFG = POINTER(PxoCfg)
F3 = POINTER(c_float)

# name -> (restype, argtypes); must list every symbol the headers under include/ declare
SIGNATURES = {
    "pxo_last_error": (c_char_p, []),
    "pxo_version": (c_int, []),
    "pxo_cfg_bytes": (c_size_t, []),
    "pxo_set_tuning": (c_int, [c_int, c_int]),
    "pxo_get_tuning": (c_int, [c_int, POINTER(c_int)]),
    "pxo_occupy_cus": (c_int, [c_int, c_int, c_float, c_int, P]),
    "pxo_tile_rows": (c_int, []),
    "pxo_param_layout": (c_int, [CFG, POINTER(PxoLeaf), POINTER(c_int64)]),
    "pxo_packed_sizes": (c_int, [CFG, POINTER(c_int64), POINTER(c_int64)]),
    "pxo_pack_weights": (c_int, [CFG, P, P, P, P]),
    "pxo_sample_along_rays": (c_int, [P, P, c_int64, c_int, c_float, c_float, c_int, P, P, P, P]),
    "pxo_posenc": (c_int, [P, c_int64, P, P]),
    "pxo_relu_mask_bytes": (c_size_t, [c_int64]),
    "pxo_mlp_fwd": (c_int, [CFG, P, P, c_int64, P, P, P, P, P, P]),
    "pxo_dbias_partial_bytes": (c_size_t, [c_int64]),
    "pxo_mlp_bwd_data": (c_int, [CFG, P, P, P, P, c_int64, P, P, P]),
    "pxo_wgrad_workspace_bytes": (c_int, [CFG, c_int64, POINTER(c_size_t)]),
    "pxo_mlp_bwd_weights": (c_int, [CFG, P, P, P, P, P, P, c_int64, P, P, c_size_t, P]),
    "pxo_shade_composite_fwd": (c_int, [CFG, P, P, P, P, P, c_int64, c_int, P, P, P, P, P]),
    "pxo_shade_composite_bwd": (c_int, [CFG, P, P, P, P, P, P, c_int64, c_int, P, P, P]),
    "pxo_shade_composite_train": (c_int, [CFG, P, P, P, P, P, P, c_int64, c_int, P, P, P, P, P, c_int64, P, P]),
    "pxo_sample_pdf": (c_int, [P, P, P, P, c_int64, c_int, c_int, P, P, P, P]),
    "pxo_add_gaussian_noise": (c_int, [P, c_int64, c_float, P, c_uint64, c_uint64, P]),
    "pxo_uniform": (c_int, [c_uint64, c_uint64, c_int64, c_float, c_float, P, P]),
    "pxo_randint": (c_int, [c_uint64, c_uint64, c_int64, c_int64, P, P]),
    "pxo_generate_rays": (c_int, [P, c_int, c_int, c_float, P, c_int64, P, P, P, P]),
    "pxo_sample_batch": (c_int, [c_uint64, c_uint64, P, c_int, c_int, c_float, P, c_int64, c_int64, P, P, P, P, P, P]),
    "pxo_generate_rays_multi": (c_int, [P, c_int, c_int, c_int, c_float, P, c_int64, P, P, P, P]),
    "pxo_mean_over_samples": (c_int, [CFG, P, P, c_int64, c_int, P, P]),
    "pxo_adam_step": (c_int, [P, P, P, P, c_int64, c_float, c_int64, c_float, P]),
    "pxo_adam_pack_step": (c_int, [CFG, P, P, P, P, c_float, c_int64, c_float, P, P, P, P, P]),
    "pxo_render_workspace_bytes": (c_int, [CFG, c_int64, POINTER(c_size_t)]),
    "pxo_render_fwd": (c_int, [CFG, P, P, P, P, P, c_int64, c_int, P, P, c_uint64, P, P, P, P, P, P, P,
                               c_size_t, P]),
    "pxo_train_workspace_bytes": (c_int, [CFG, c_int64, POINTER(c_size_t)]),
    "pxo_train_fwd_bwd": (c_int, [CFG, P, P, P, P, P, P, P, P, P, c_int64, c_int, P, P, P, c_uint64, P, P,
                                  P, c_size_t, P]),
    "pxo_train_fwd_bwd_bucketed": (c_int, [CFG, P, P, P, P, P, P, P, P, P, c_int64, c_int, P, P, P, c_uint64, P, P,
                                           P, c_size_t, P, P]),
    "pxo_train_backward_work": (c_int, [CFG, c_int64, P, c_size_t, POINTER(c_int64), POINTER(c_int64), P]),
    "pxo_event_create": (c_int, [POINTER(c_void_p)]),
    "pxo_event_destroy": (c_int, [P]),
    "pxo_stream_wait_event": (c_int, [P, P]),
    "pxo_eval_points": (c_int, [CFG, P, P, c_int64, P, P, P]),
    "pxo_profile_enable": (c_int, [c_int]),
    "pxo_profile_read": (c_int, [c_int, POINTER(c_int64), POINTER(ctypes.c_double), POINTER(c_int64)]),
    "pxo_grid_sigma": (c_int, [CFG, P, c_int, c_int, c_int, POINTER(c_float), POINTER(c_float), P, P]),
    # include/plenoctree_octree.h
    "pxo_threshold_mask": (c_int, [P, c_int64, c_float, P, P]),
    "pxo_tree_workspace_bytes": (c_int, [c_int, POINTER(c_size_t)]),
    "pxo_tree_count_nodes": (c_int, [P, c_int, P, c_size_t, POINTER(c_int64), P]),
    "pxo_tree_build": (c_int, [P, c_size_t, c_int, POINTER(c_int64), P, P, P]),
    "pxo_tree_sample_cells": (c_int, [P, c_int64, c_int64, c_int, P, F3, F3, P, P]),
    "pxo_tree_sample_leaves": (c_int, [P, P, c_int64, c_int, P, F3, F3, P, P]),
    "pxo_tree_query": (c_int, [P, P, c_int64, F3, F3, P, P]),
    "pxo_tree_relu_sigma": (c_int, [P, c_int64, c_int, P]),
    "pxo_grid_weight_workspace_bytes": (c_int, [c_int, POINTER(c_size_t)]),
    "pxo_grid_weight_render": (c_int, [P, c_int, P, c_int, c_float, c_float, c_int, c_int, POINTER(PxoRenderOpts),
                                       F3, F3, P, P, c_size_t, P]),
    "pxo_octree_set_lanes_per_ray": (c_int, [c_int, c_int]),
    "pxo_octree_set_tuning": (c_int, [c_int, c_int]),
    "pxo_octree_get_tuning": (c_int, [c_int, POINTER(c_int)]),
    "pxo_octree_render_fwd": (c_int, [POINTER(PxoTree), POINTER(PxoCamera), P, P, P, c_int64,
                                      POINTER(PxoRenderOpts), P, P]),
    "pxo_octree_render_bwd": (c_int, [POINTER(PxoTree), POINTER(PxoCamera), P, P, P, c_int64,
                                      POINTER(PxoRenderOpts), P, P, P, P]),
    "pxo_octree_count_work": (c_int, [POINTER(PxoTree), POINTER(PxoCamera), POINTER(PxoRenderOpts), P, P, P]),
    "pxo_grid_weight_count_work": (c_int, [P, c_int, P, c_int, c_float, c_float, c_int, c_int, POINTER(PxoRenderOpts),
                                           F3, F3, P, P, P]),
    "pxo_image_mse": (c_int, [P, P, c_int64, P, P, P]),
    "pxo_sgd_step": (c_int, [P, P, P, c_int64, c_float, c_float, c_int, c_int, P]),
}

_lib = None


def load(path=None):
    """dlopen the in-tree library; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if path is None and os.environ.get("PXO_LIB"):
        # a variant library (kernel A/B sessions, scripts/) is only honoured when explicitly allowed: nothing that
        # is benchmarked or tested by default can pick up anything but the in-tree build
        if os.environ.get("PXO_ALLOW_VARIANT") != "1":
            raise PxoError("PXO_LIB is set but PXO_ALLOW_VARIANT != 1: refusing to load a variant kernel library")
        path = os.environ["PXO_LIB"]
    path = path or LIB_PATH
    # torch must load (and initialise) its HIP runtime first: the library then binds to the same
    # libamdhip64 instance, so device pointers and streams are shared with torch.
    import torch
    if torch.cuda.is_available():
        torch.cuda.init()
    if not os.path.exists(path):
        raise PxoError(
            f"{path} not found: build it with `python -m plenoctree_amd.build` "
            "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    lib = ctypes.CDLL(path)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = restype
        fn.argtypes = argtypes
    # the struct layouts of this module are those of include/plenoctree_hip.h at ABI_VERSION: a stale library (or a stale
    # copy of this file) would pass a short PxoCfg and have its tail read from past the end
    if lib.pxo_version() != ABI_VERSION or lib.pxo_cfg_bytes() != ctypes.sizeof(PxoCfg):
        raise PxoError(f"{path}: ABI version {lib.pxo_version()} / sizeof(PxoCfg) {lib.pxo_cfg_bytes()}, this binding expects "
                       f"{ABI_VERSION} / {ctypes.sizeof(PxoCfg)}: rebuild with `python -m plenoctree_amd.build`")
    _lib = lib
    return lib


def check(status, what):
    if status != 0:
        msg = load().pxo_last_error()
        raise PxoError(f"{what} failed ({status}): {msg.decode() if msg else ''}")


def make_cfg(**kw):
    """PxoCfg with the defaults of nerf_sh/config/blender.yaml over nerf_sh/nerf/utils.py:61-230."""
    vals = dict(num_coarse_samples=64, num_fine_samples=128, sh_deg=3, min_deg_point=0, max_deg_point=10,
                white_bkgd=1, lindisp=0, sparsity_npoints=10000, near_=2.0, far_=6.0,
                sparsity_weight=1e-3, sparsity_length=0.05, sparsity_radius=1.5, weight_decay_mult=0.0,
                mlp_precision=0, noise_std=0.0, skip_zero_rows=0)
    for k, v in kw.items():
        if k not in vals:
            raise ValueError(f"unknown PxoCfg field {k}")
        vals[k] = v
    cfg = PxoCfg()
    for k, v in vals.items():
        setattr(cfg, k, int(v) if k not in ("near_", "far_", "sparsity_weight", "sparsity_length",
                                            "sparsity_radius", "weight_decay_mult", "noise_std") else float(v))
    return cfg
