"""torch-tensor front end of include/plenoctree_octree.h (tree build, weight mask, octree renderer).

As in ops.py, torch only provides device storage and the current HIP stream; every operation runs in
libplenoctree_hip.so and there is no CPU fallback.
"""
import ctypes

import torch

from . import _lib
from ._lib import PxoError, check
from .ops import _f, _new, _p, _require_gpu, _stream


def _vec3(v):
    return (ctypes.c_float * 3)(*[float(x) for x in v])


def render_opts(step_size=1e-3, background_brightness=1.0, sigma_thresh=0.0, stop_thresh=0.0):
    return _lib.PxoRenderOpts(float(step_size), float(background_brightness), float(sigma_thresh), float(stop_thresh))


def threshold_mask(value, thresh):
    """uint8 mask = value >= thresh (octree/extraction.py:322-331)."""
    _require_gpu()
    value = value.reshape(-1)
    mask = _new(value.numel(), device=value.device, dtype=torch.uint8)
    check(_lib.load().pxo_threshold_mask(_f(value), value.numel(), float(thresh), _p(mask), _stream()),
          "pxo_threshold_mask")
    return mask


def tree_workspace_bytes(depth):
    n = ctypes.c_size_t(0)
    check(_lib.load().pxo_tree_workspace_bytes(depth, ctypes.byref(n)), "pxo_tree_workspace_bytes")
    return n.value


def tree_from_mask(mask, depth):
    """Octree of the masked voxels of a 2^(depth+1) grid: (child [n,2,2,2] int32, parent_depth [n,2] int32,
    level_nodes list).  Equivalent of `depth` rounds of tree[grid].refine() (octree/extraction.py:341-350)."""
    _require_gpu()
    lib = _lib.load()
    reso = 2 ** (depth + 1)
    if mask.dtype != torch.uint8 or mask.numel() != reso ** 3:
        raise PxoError(f"mask must be uint8 with {reso}^3 entries")
    nbytes = tree_workspace_bytes(depth)
    ws = _new(nbytes, device=mask.device, dtype=torch.uint8)
    levels = (ctypes.c_int64 * (depth + 1))()
    check(lib.pxo_tree_count_nodes(_p(mask.reshape(-1)), depth, _p(ws), nbytes, levels, _stream()),
          "pxo_tree_count_nodes")
    level_nodes = [int(v) for v in levels]
    n = sum(level_nodes)
    child = _new(n, 2, 2, 2, device=mask.device, dtype=torch.int32)
    parent_depth = _new(n, 2, device=mask.device, dtype=torch.int32)
    check(lib.pxo_tree_build(_p(ws), nbytes, depth, levels, _p(child), _p(parent_depth), _stream()), "pxo_tree_build")
    return child, parent_depth, level_nodes


def tree_sample_cells(parent_depth, node0, n_nodes, samples, offset, invradius, u=None, seed=0, stream_id=0):
    """World-space sample points [n_nodes*8, samples, 3] of the 8 cells of nodes [node0, node0+n_nodes)."""
    _require_gpu()
    lib = _lib.load()
    n = n_nodes * 8 * samples
    dev = parent_depth.device
    if u is None:
        u = _new(max(n * 3, 1), device=dev)
        check(lib.pxo_uniform(seed, stream_id, n * 3, 0.0, 1.0, _f(u), _stream()), "pxo_uniform")
    pts = _new(n_nodes * 8, samples, 3, device=dev)
    check(lib.pxo_tree_sample_cells(_p(parent_depth), node0, n_nodes, samples, _f(u), _vec3(offset), _vec3(invradius),
                                    _f(pts), _stream()), "pxo_tree_sample_cells")
    return pts


def tree_sample_leaves(parent_depth, packed, samples, offset, invradius, u=None, seed=0, stream_id=0):
    """World-space sample points [n, samples, 3] inside the leaves `packed` (int64 node*8+cell)."""
    _require_gpu()
    lib = _lib.load()
    if packed.dtype != torch.int64:
        raise PxoError("packed leaf indices must be int64")
    n_cells = packed.numel()
    n = n_cells * samples
    dev = parent_depth.device
    if u is None:
        u = _new(max(n * 3, 1), device=dev)
        check(lib.pxo_uniform(seed, stream_id, n * 3, 0.0, 1.0, _f(u), _stream()), "pxo_uniform")
    pts = _new(n_cells, samples, 3, device=dev)
    check(lib.pxo_tree_sample_leaves(_p(parent_depth), _p(packed.contiguous()), n_cells, samples, _f(u), _vec3(offset),
                                     _vec3(invradius), _f(pts), _stream()), "pxo_tree_sample_leaves")
    return pts


def tree_query(child, points, offset, invradius):
    """Packed index (node*8 + cell, int64 [n]) of the leaf containing each world-space point [n,3]."""
    _require_gpu()
    points = points.reshape(-1, 3).contiguous().float()
    out = _new(points.shape[0], device=points.device, dtype=torch.int64)
    check(_lib.load().pxo_tree_query(_p(child), _f(points), points.shape[0], _vec3(offset), _vec3(invradius), _p(out),
                                     _stream()), "pxo_tree_query")
    return out


def tree_relu_sigma(data):
    _require_gpu()
    dim = data.shape[-1]
    check(_lib.load().pxo_tree_relu_sigma(_f(data), data.numel() // dim, dim, _stream()), "pxo_tree_relu_sigma")


def grid_weight_render(sigma_grid, reso, c2w_all, fx, fy, width, height, opts, offset, invradius, grid_weight=None):
    """Per-voxel maximum compositing weight over all pixels of the given cameras (c2w_all [n,3,4] or [n,4,4])."""
    _require_gpu()
    lib = _lib.load()
    dev = sigma_grid.device
    c2w = c2w_all[:, :3, :4].contiguous().to(device=dev, dtype=torch.float32)
    if grid_weight is None:
        grid_weight = torch.zeros(reso ** 3, dtype=torch.float32, device=dev)
    nbytes = ctypes.c_size_t(0)
    check(lib.pxo_grid_weight_workspace_bytes(reso, ctypes.byref(nbytes)), "pxo_grid_weight_workspace_bytes")
    ws = _new(max(nbytes.value, 16), device=dev, dtype=torch.uint8)
    check(lib.pxo_grid_weight_render(_f(sigma_grid.reshape(-1)), reso, _f(c2w), c2w.shape[0], float(fx), float(fy),
                                     int(width), int(height), ctypes.byref(opts), _vec3(offset), _vec3(invradius),
                                     _f(grid_weight), _p(ws), nbytes.value, _stream()), "pxo_grid_weight_render")
    return grid_weight


def tree_view(child, data, offset, invradius):
    """PxoTree struct over torch storage (keeps no reference: the caller owns the tensors)."""
    data_dim = data.shape[-1]
    if (data_dim - 1) % 3:
        raise PxoError(f"data_dim {data_dim} is not 3*basis_dim+1")
    if child.dtype != torch.int32:
        raise PxoError("child must be int32")
    t = _lib.PxoTree()
    t.child = _p(child).value
    t.data = _f(data).value
    t.n_internal = child.shape[0]
    t.data_dim = data_dim
    t.basis_dim = (data_dim - 1) // 3
    t.offset = _vec3(offset)
    t.invradius = _vec3(invradius)
    return t


def _camera(c2w, width, height, fx, fy):
    c2w = c2w[:3, :4].contiguous().float()
    cam = _lib.PxoCamera(_f(c2w).value, float(fx), float(fx if fy is None else fy), int(width), int(height))
    return cam, c2w          # keep c2w alive for the duration of the call


def set_lanes_per_ray(forward=0, backward=0):
    """Lanes per ray of the renderer launches (4 / 8 / 16; 0 = measured default 4)."""
    check(_lib.load().pxo_octree_set_lanes_per_ray(int(forward), int(backward)), "pxo_octree_set_lanes_per_ray")


TUNE_GW_MARCHER, TUNE_BWD_CACHE_ROWS, TUNE_BWD_UPDATE, TUNE_GW_TILE_ORDER = 0, 1, 2, 3


def get_tuning(knob):
    v = ctypes.c_int(0)
    check(_lib.load().pxo_octree_get_tuning(int(knob), ctypes.byref(v)), "pxo_octree_get_tuning")
    return v.value


def set_tuning(knob, value):
    """pxo_octree_set_tuning: choose between kernels that compute the same result (A/B sessions, equality tests)."""
    check(_lib.load().pxo_octree_set_tuning(int(knob), int(value)), "pxo_octree_set_tuning")


def octree_render_persp(tree, c2w, width, height, fx, opts, fy=None):
    """[H,W,3] image of a pinhole camera (VolumeRenderer.render_persp)."""
    _require_gpu()
    cam, keep = _camera(c2w, width, height, fx, fy)
    out = _new(height, width, 3, device=keep.device)
    check(_lib.load().pxo_octree_render_fwd(ctypes.byref(tree), ctypes.byref(cam), None, None, None, width * height,
                                            ctypes.byref(opts), _f(out), _stream()), "pxo_octree_render_fwd")
    return out


def octree_render_persp_bwd(tree, c2w, width, height, fx, opts, grad_out, grad_data, fy=None, out_rgb=None):
    """Accumulates d sum(image * grad_out) / d data into grad_data; `out_rgb` = the exact forward image of the same
    camera (saves one of the two marches) or None."""
    _require_gpu()
    cam, keep = _camera(c2w, width, height, fx, fy)
    check(_lib.load().pxo_octree_render_bwd(ctypes.byref(tree), ctypes.byref(cam), None, None, None, width * height,
                                            ctypes.byref(opts), _f(out_rgb), _f(grad_out), _f(grad_data), _stream()),
          "pxo_octree_render_bwd")
    return grad_data


def octree_count_work(tree, c2w, width, height, fx, opts, fy=None, count_leaves=True):
    """Work counters of one render of `tree` from camera c2w (pxo_octree_count_work: the renderer's own march, counting):
    dict(rays, samples, shaded_samples, child_loads, distinct_leaves).  A roofline / debug pass, not on the product path."""
    _require_gpu()
    cam, keep = _camera(c2w, width, height, fx, fy)
    counts = torch.zeros(4, dtype=torch.int64, device=keep.device)
    seen = torch.zeros(tree.n_internal * 8, dtype=torch.uint8, device=keep.device) if count_leaves else None
    check(_lib.load().pxo_octree_count_work(ctypes.byref(tree), ctypes.byref(cam), ctypes.byref(opts), _p(counts), _p(seen),
                                            _stream()), "pxo_octree_count_work")
    c = counts.tolist()
    return {"rays": c[0], "samples": c[1], "shaded_samples": c[2], "child_loads": c[3],
            "distinct_leaves": int(seen.sum(dtype=torch.int64)) if count_leaves else None}


def grid_weight_count_work(sigma_grid, reso, c2w_all, fx, fy, width, height, opts, offset, invradius, count_voxels=True):
    """Work counters of pxo_grid_weight_render on the same cameras: dict(rays, samples, occupied_samples, distinct_voxels)."""
    _require_gpu()
    dev = sigma_grid.device
    c2w = c2w_all[:, :3, :4].contiguous().to(device=dev, dtype=torch.float32)
    counts = torch.zeros(4, dtype=torch.int64, device=dev)
    seen = torch.zeros(reso ** 3, dtype=torch.uint8, device=dev) if count_voxels else None
    check(_lib.load().pxo_grid_weight_count_work(_f(sigma_grid.reshape(-1)), reso, _f(c2w), c2w.shape[0], float(fx), float(fy),
                                                 int(width), int(height), ctypes.byref(opts), _vec3(offset), _vec3(invradius),
                                                 _p(counts), _p(seen), _stream()), "pxo_grid_weight_count_work")
    c = counts.tolist()
    return {"rays": c[0], "samples": c[1], "occupied_samples": c[2],
            "distinct_voxels": int(seen.sum(dtype=torch.int64)) if count_voxels else None}


def octree_render_rays(tree, origins, dirs, viewdirs, opts):
    """[B,3] colours of explicit world-space rays with unit dirs (VolumeRenderer.forward)."""
    _require_gpu()
    B = origins.shape[0]
    out = _new(B, 3, device=origins.device)
    check(_lib.load().pxo_octree_render_fwd(ctypes.byref(tree), None, _f(origins), _f(dirs), _f(viewdirs), B,
                                            ctypes.byref(opts), _f(out), _stream()), "pxo_octree_render_fwd")
    return out


def octree_render_rays_bwd(tree, origins, dirs, viewdirs, opts, grad_out, grad_data, out_rgb=None):
    _require_gpu()
    check(_lib.load().pxo_octree_render_bwd(ctypes.byref(tree), None, _f(origins), _f(dirs), _f(viewdirs),
                                            origins.shape[0], ctypes.byref(opts), _f(out_rgb), _f(grad_out), _f(grad_data),
                                            _stream()), "pxo_octree_render_bwd")
    return grad_data


def image_mse(im, gt, want_grad=True):
    """(sse device scalar, grad or None) of mean((clamp(im,0,1)-gt)^2)  (octree/optimization.py:217-219)."""
    _require_gpu()
    n = im.numel()
    grad = torch.empty_like(im) if want_grad else None
    sse = _new(1, device=im.device)
    check(_lib.load().pxo_image_mse(_f(im), _f(gt), n, _f(grad), _f(sse), _stream()), "pxo_image_mse")
    return sse, grad


def sgd_step(params, grads, lr, momentum=0.0, nesterov=False, buf=None, first_step=False):
    _require_gpu()
    check(_lib.load().pxo_sgd_step(_f(params), _f(grads), _f(buf), params.numel(), float(lr), float(momentum), int(nesterov),
                                   int(first_step), _stream()), "pxo_sgd_step")
