"""Host-side mirror of the slice of `svox` the reference uses (N3Tree, VolumeRenderer, helpers) on the MI355X path.

svox (pin svox>=0.2.28) is a third-party dependency of the reference and is not part of its tree; the classes
below keep its names, argument meaning, indexing surface and npz format for exactly the calls the reference makes,
so that the reference's own octree/extraction.py (step1, step2), octree/optimization.py and octree/evaluation.py
run UNCHANGED with `sys.modules["svox"]` pointing here (tests/test_reference_drivers_cpu.py does exactly that):

    N3Tree(N, data_dim, init_refine=0, init_reserve, geom_resize_fact, depth_limit, radius, center, data_format,
           extra_data, map_location)                                               octree/extraction.py:489-499
    tree.offset.cpu(), tree.invradius.cpu()                                        octree/extraction.py:291-292
    tree[grid].refine()         (grid: [n,3] world-space points)                   octree/extraction.py:341-350
    tree.max_depth, tree.depths, tree[leaf_inds].sample(S), tree[leaf_inds] = rgba octree/extraction.py:353-394
    tree.data_format.format == tree.data_format.RGBA, tree.data_dim                octree/extraction.py:377-378
    tree[:, -1:].relu_(), tree.shrink_to_fit(), print(tree), tree.save(path, compress=False)   :503-509
    svox.N3Tree.load(path, map_location), t.parameters(), t.data.dtype / .grad / .device, t.clone(device='cpu')
                                                                                   octree/optimization.py:167-240
    svox.VolumeRenderer(t, step_size, ndc).render_persp(c2w, height, width, fx, fast, cuda)
                                                                  octree/nerf/utils.py:456-474, octree/optimization.py:174-216
    svox.NDCConfig(width, height, focal)                          (constructed only for LLFF configs: not supported)
    svox.helpers._get_c_extension(): RenderOptions(), CameraSpec(), grid_weight_render(...)    octree/extraction.py:181-214

Two ways to build a tree: the generic `tree[points].refine()` above (leaf lookup in HIP, structure edits as tensor
index operations, as svox itself does them), and `refine_from_mask` -- the whole init_grid_depth refinement of the
masked grid in one pass (Morton pyramid + scans, 0.3 ms at 512^3), which our own extraction driver uses; with whole
levels refined per call both produce the same arrays (tests/test_gpu_octree.py).

Storage and node order are svox's (include/plenoctree_octree.h).  Rendering, sampling, leaf lookup, the weight mask
and the tree build run in libplenoctree_hip.so through plenoctree_amd.octree_ops; torch holds the device arrays.  Only
N = 2 and the SH data formats (`SH1/4/9/16/25`) are supported -- anything else raises, there is no fallback.
"""
import re
import types

import numpy as np
import torch

from .. import octree_ops as oops
from .._lib import PxoError, TREE_MAX_DEPTH


def _vec3(v, name):
    a = np.asarray(v, np.float32).reshape(-1)
    if a.size == 1:
        a = np.repeat(a, 3)
    if a.size != 3:
        raise ValueError(f"{name} must have 1 or 3 entries")
    return a.astype(np.float32)


class DataFormat:
    """svox.DataFormat: `.format` in {RGBA, SH, SG, ASG}, `.basis_dim`; str() gives the npz spelling ('SH16')."""
    RGBA, SH, SG, ASG = 0, 1, 2, 3

    def __init__(self, txt, data_dim=None):
        m = re.fullmatch(r"SH(\d+)", str(txt))
        if not m:
            raise NotImplementedError(f"data_format {txt!r}: only the spherical-harmonics formats SH1/4/9/16/25 are supported")
        k = int(m.group(1))
        if k not in (1, 4, 9, 16, 25):
            raise ValueError(f"data_format {txt}: basis_dim must be a square <= 25")
        if data_dim is not None and data_dim != 3 * k + 1:
            raise ValueError(f"data_dim {data_dim} does not match data_format {txt} (expected {3 * k + 1})")
        self.format = DataFormat.SH
        self.basis_dim = k
        self.data_dim = 3 * k + 1

    def __repr__(self):
        return f"SH{self.basis_dim}"

    __str__ = __repr__

    def __eq__(self, other):
        return str(self) == str(other)

    def __hash__(self):
        return hash(str(self))


def parse_data_format(fmt, data_dim):
    """'SH16' -> basis_dim 16."""
    return DataFormat(fmt, data_dim).basis_dim


class NDCConfig:
    """svox.NDCConfig (forward-facing LLFF scenes).  Constructible, but the renderer rejects it: NDC is out of scope."""

    def __init__(self, width, height, focal, near=1.0):
        self.width, self.height, self.focal, self.near = width, height, focal, near


class N3TreeView:
    """`tree[key]`: a set of leaves (given by packed cell index node*8 + cell) and a channel slice."""

    def __init__(self, tree, packed, channels=slice(None)):
        self.tree, self.packed, self.channels = tree, packed, channels

    def _all(self):
        return self.packed is None

    def _packed(self):
        return self.tree._leaf_packed() if self.packed is None else self.packed

    def refine(self, repeats=1):
        """svox N3TreeView.refine: split every selected leaf (unique) into 2^3 children holding its value.  Returns
        the number of nodes added.  `repeats` > 1 re-selects all leaves each time and is only defined for tree[:]."""
        if repeats != 1 and not self._all():
            raise NotImplementedError("refine(repeats > 1) is only supported on tree[:]")
        return sum(self.tree._refine_packed(self._packed()) for _ in range(repeats))

    def sample(self, n_samples, device=None):
        """[n_leaves, n_samples, 3] uniform world-space points inside each selected leaf (octree/extraction.py:369)."""
        t = self.tree
        t._sample_calls += 1
        return oops.tree_sample_leaves(t.parent_depth, self._packed(), n_samples, t.offset, t.invradius,
                                       seed=t._sample_seed, stream_id=0x51A0 + t._sample_calls)

    @property
    def values(self):
        d = self.tree.data.data.view(-1, self.tree.data_dim)
        return d[:, self.channels] if self._all() else d[self.packed][:, self.channels]

    @property
    def depths(self):
        return self.tree.parent_depth[self._packed() >> 3, 1]

    def relu_(self):
        """`tree[:, -1:].relu_()` (octree/extraction.py:503) runs the HIP kernel; other selections index tensors."""
        t = self.tree
        ch = range(t.data_dim)[self.channels]
        if self._all() and list(ch) == [t.data_dim - 1]:
            oops.tree_relu_sigma(t.data.data)
            return self
        d = t.data.data.view(-1, t.data_dim)
        with torch.no_grad():
            if self._all():
                d[:, self.channels] = d[:, self.channels].relu()
            else:
                rows = d[self.packed]
                rows[:, self.channels] = rows[:, self.channels].relu()
                d[self.packed] = rows
        return self

    def set(self, value):
        t = self.tree
        d = t.data.data.view(-1, t.data_dim)
        value = torch.as_tensor(value, dtype=d.dtype, device=d.device)
        with torch.no_grad():
            if self._all():
                d[:, self.channels] = value
            elif self.channels == slice(None):
                d[self.packed] = value
            else:
                rows = d[self.packed]
                rows[:, self.channels] = value
                d[self.packed] = rows


class N3Tree:
    def __init__(self, N=2, data_dim=None, depth_limit=10, init_reserve=1, init_refine=0, geom_resize_fact=1.0,
                 radius=0.5, center=(0.5, 0.5, 0.5), data_format="RGBA", extra_data=None, device="cpu",
                 map_location=None):
        if N != 2:
            raise NotImplementedError("only octrees (tree_branch_n = 2) are supported")
        if init_refine != 0:
            raise NotImplementedError("init_refine must be 0 (octree/extraction.py:491)")
        if extra_data is not None:
            raise NotImplementedError("extra_data (spherical-gaussian formats) is not supported")
        if not 1 <= depth_limit <= TREE_MAX_DEPTH:
            raise ValueError(f"depth_limit must be in [1, {TREE_MAX_DEPTH}]")
        self.N = 2
        self.data_format = DataFormat(data_format, data_dim)
        self.depth_limit = int(depth_limit)
        self.geom_resize_fact = float(geom_resize_fact)
        dev = torch.device(map_location if map_location is not None else device)
        radius, center = _vec3(radius, "radius"), _vec3(center, "center")
        self._set_transform((np.float32(0.5) / radius).astype(np.float32),
                            (np.float32(0.5) * (np.float32(1.0) - center / radius)).astype(np.float32))
        self.child = torch.zeros(1, 2, 2, 2, dtype=torch.int32, device=dev)
        self.parent_depth = torch.zeros(1, 2, dtype=torch.int32, device=dev)
        self.data = torch.nn.Parameter(torch.zeros(1, 2, 2, 2, self.data_dim, dtype=torch.float32, device=dev))
        self.level_nodes = [1]
        self._reset_caches()

    def _set_transform(self, invradius, offset):
        # host float32 tensors (svox keeps them as buffers; the reference calls .cpu() on them and hands them back)
        self.invradius = torch.from_numpy(np.ascontiguousarray(invradius, dtype=np.float32))
        self.offset = torch.from_numpy(np.ascontiguousarray(offset, dtype=np.float32))

    def _reset_caches(self):
        self._leaves = None            # packed indices of all leaves, svox order
        self._level_order = True       # nodes stored breadth-first, levels contiguous (fast-path methods need it)
        self._sample_seed, self._sample_calls = 0, 0

    # ---- structure -------------------------------------------------------------------------------
    @property
    def basis_dim(self):
        return self.data_format.basis_dim

    @property
    def data_dim(self):
        return self.data_format.data_dim

    @property
    def device(self):
        return self.data.device

    @property
    def n_internal(self):
        return self.child.shape[0]

    @property
    def capacity(self):
        return self.child.shape[0]

    @property
    def n_leaves(self):
        return 7 * self.n_internal + 1          # every internal node replaces one leaf by eight

    @property
    def max_depth(self):
        return len(self.level_nodes) - 1

    @property
    def depths(self):
        """Depth of every leaf (= depth of the node that holds it), leaves in svox order (octree/extraction.py:358)."""
        return self.parent_depth[self._leaf_packed() >> 3, 1]

    def parameters(self):
        """nn.Module.parameters() of svox's N3Tree: the data array (octree/optimization.py:180-186)."""
        return [self.data]

    def __repr__(self):
        return (f"svox.N3Tree(N={self.N}, data_dim={self.data_dim}, depth_limit={self.depth_limit}, "
                f"capacity:{self.n_internal}/{self.capacity}, data_format:{self.data_format})")

    def _leaf_packed(self):
        """int64 packed index node*8 + cell of every leaf, ascending = svox's leaf order (nonzero of child == 0)."""
        if self._leaves is None:
            self._leaves = torch.nonzero(self.child.view(-1) == 0).reshape(-1)
        return self._leaves

    # ---- indexing (svox N3Tree.__getitem__ / __setitem__ for the reference's uses) ---------------------
    def _view(self, key):
        channels = slice(None)
        if isinstance(key, tuple):
            if len(key) != 2:
                raise NotImplementedError("tree[...] takes a leaf selector and optionally a channel slice")
            key, channels = key
            if isinstance(channels, int):
                channels = slice(channels, channels + 1 if channels != -1 else None)
            if not isinstance(channels, slice):
                raise NotImplementedError("the channel selector must be an int or a slice")
        if isinstance(key, slice):
            if key != slice(None):
                raise NotImplementedError("leaf slices other than ':' are not supported")
            return N3TreeView(self, None, channels)
        key = torch.as_tensor(key, device=self.device)
        if key.is_floating_point():
            if key.shape[-1] != 3:
                raise ValueError("tree[points]: points must be [n, 3] world coordinates")
            return N3TreeView(self, oops.tree_query(self.child, key.to(self.device), self.offset, self.invradius), channels)
        if key.dtype == torch.bool:
            key = torch.nonzero(key.reshape(-1)).reshape(-1)
        return N3TreeView(self, self._leaf_packed()[key.reshape(-1).long()], channels)

    def __getitem__(self, key):
        return self._view(key)

    def __setitem__(self, key, value):
        self._view(key).set(value)

    def _refine_packed(self, packed):
        """Splits the leaves `packed` (any order, duplicates allowed): new nodes are appended in ascending packed
        order of their parent cell (svox N3Tree._refine_at / refine), children inherit the leaf's value."""
        with torch.no_grad():
            packed = torch.unique(packed.reshape(-1))
            node = packed >> 3
            depth = self.parent_depth[node, 1]
            keep = depth < self.depth_limit
            packed, node, depth = packed[keep], node[keep], depth[keep]
            k = int(packed.numel())
            if k == 0:
                return 0
            n0, D, dev = self.n_internal, self.data_dim, self.device
            if n0 + k >= 1 << 28:
                raise PxoError(f"{n0 + k} nodes exceed the int32 packed-index range")
            if int(depth.min()) < self.max_depth:
                self._level_order = False          # a shallower leaf split after deeper nodes exist
            child = torch.cat([self.child, torch.zeros(k, 2, 2, 2, dtype=torch.int32, device=dev)])
            new = torch.arange(n0, n0 + k, device=dev)
            child.view(-1)[packed] = (new - node).int()
            old = self.data.data
            data = torch.cat([old, old.view(-1, D)[packed][:, None, :].expand(k, 8, D).reshape(k, 2, 2, 2, D)])
            pd = torch.cat([self.parent_depth, torch.stack([packed.int(), depth + 1], 1)])
            self.child, self.parent_depth = child, pd
            self.data = torch.nn.Parameter(data, requires_grad=self.data.requires_grad)
            self.level_nodes = [int(c) for c in torch.bincount(pd[:, 1].long()).tolist()]
            self._leaves = None
            return k

    def refine_from_mask(self, mask):
        """Replaces `for _ in range(depth): tree[grid].refine()` (octree/extraction.py:337-350), where grid are
        the centres of the masked voxels of the 2^(depth_limit+1) grid: `mask` is that grid's uint8 mask
        (x slowest).  Must be called on a fresh tree (root only)."""
        if self.n_internal != 1:
            raise PxoError("refine_from_mask needs a fresh tree")
        self.child, self.parent_depth, self.level_nodes = oops.tree_from_mask(mask.reshape(-1), self.depth_limit)
        while len(self.level_nodes) > 1 and self.level_nodes[-1] == 0:
            self.level_nodes.pop()
        self.data = torch.nn.Parameter(torch.zeros(self.n_internal, 2, 2, 2, self.data_dim, dtype=torch.float32,
                                                   device=mask.device), requires_grad=self.data.requires_grad)
        self._leaves, self._level_order = None, True
        return self

    def _need_level_order(self, what):
        if not self._level_order:
            raise PxoError(f"{what} needs breadth-first node storage (levels refined one at a time)")

    def max_depth_nodes(self):
        """(first node, count) of the deepest level: its 8*count cells are the leaves with depth == max_depth,
        in leaf order (`tree.depths == tree.max_depth`, octree/extraction.py:358-360)."""
        self._need_level_order("max_depth_nodes")
        count = self.level_nodes[-1]
        return self.n_internal - count, count

    def sample_max_depth_cells(self, n_samples, first=0, count=None, seed=0, u=None):
        """tree[leaf_ind[...]].sample(n_samples) for the cells of `count` deepest-level nodes starting at the
        `first`-th: [count*8, n_samples, 3] world-space points (octree/extraction.py:369)."""
        node0, total = self.max_depth_nodes()
        count = total - first if count is None else count
        return oops.tree_sample_cells(self.parent_depth, node0 + first, count, n_samples, self.offset, self.invradius,
                                      u=u, seed=seed, stream_id=0x7A3 + first)

    def max_depth_data(self, first=0, count=None):
        """View [count*8, data_dim] of the deepest-level cells: writing it is `tree[chunk_inds] = rgba`
        (octree/extraction.py:394)."""
        node0, total = self.max_depth_nodes()
        count = total - first if count is None else count
        return self.data.data[node0 + first: node0 + first + count].view(count * 8, self.data_dim)

    def relu_sigma_(self):
        """tree[:, -1:].relu_() (octree/extraction.py:503)."""
        oops.tree_relu_sigma(self.data.data)
        return self

    def shrink_to_fit(self):
        return self                              # arrays are always exactly n_internal long

    def view(self):
        return oops.tree_view(self.child, self.data.data, self.offset, self.invradius)

    def clone(self, device=None):
        t = object.__new__(N3Tree)
        t.__dict__.update(self.__dict__)
        dev = torch.device(device) if device is not None else self.device
        t.child, t.parent_depth = (x.detach().clone().to(dev) for x in (self.child, self.parent_depth))
        t.data = torch.nn.Parameter(self.data.data.detach().clone().to(dev), requires_grad=self.data.requires_grad)
        t.invradius, t.offset = self.invradius.clone(), self.offset.clone()
        t.level_nodes = list(self.level_nodes)
        t._leaves = None
        return t

    # ---- npz (svox N3Tree.save / load; keys as consumed by octree/compression.py:76-86) ------------------
    def save(self, path, shrink=True, compress=True):
        z = {
            "data_dim": self.data_dim,
            "child": self.child.cpu().numpy(),
            "parent_depth": self.parent_depth.cpu().numpy(),
            "n_internal": self.n_internal,
            "n_free": 0,
            "invradius3": self.invradius.numpy(),
            "offset": self.offset.numpy(),
            "depth_limit": self.depth_limit,
            "geom_resize_fact": self.geom_resize_fact,
            "data": self.data.data.detach().half().cpu().numpy(),       # svox stores float16
            "data_format": str(self.data_format),
        }
        (np.savez_compressed if compress else np.savez)(path, **z)

    @classmethod
    def load(cls, path, device="cpu", map_location=None):
        dev = torch.device(map_location if map_location is not None else device)
        z = np.load(path)
        if "quant_colors" in z.files:
            z = _dequantized(z)
        t = object.__new__(cls)
        t.N = 2
        fmt = str(z["data_format"]) if "data_format" in z.files else "RGBA"
        t.data_format = DataFormat(fmt, int(z["data_dim"]))
        child = z["child"]
        if child.shape[1:] != (2, 2, 2):
            raise NotImplementedError("only octrees (N = 2) are supported")
        n = int(z["n_internal"]) if "n_internal" in z.files else child.shape[0]
        t.child = torch.from_numpy(child[:n].astype(np.int32)).to(dev)
        t.data = torch.nn.Parameter(torch.from_numpy(z["data"][:n].astype(np.float32)).to(dev))
        if "parent_depth" in z.files:
            pd = z["parent_depth"][:n].astype(np.int32)
        else:
            pd = parent_depth_from_child(child[:n])
        t.parent_depth = torch.from_numpy(pd).to(dev)
        if "invradius3" in z.files:
            invradius = z["invradius3"].astype(np.float32)
        else:
            invradius = np.repeat(np.float32(z["invradius"]), 3)
        t._set_transform(invradius, z["offset"].astype(np.float32))
        t.depth_limit = int(z["depth_limit"]) if "depth_limit" in z.files else int(pd[:, 1].max())
        t.geom_resize_fact = float(z["geom_resize_fact"]) if "geom_resize_fact" in z.files else 1.0
        counts = np.bincount(pd[:, 1])
        t._reset_caches()
        t._level_order = not bool((np.diff(pd[:, 1]) < 0).any())
        t.level_nodes = [int(c) for c in counts]
        if t.max_depth > TREE_MAX_DEPTH:
            raise NotImplementedError(f"tree depth {t.max_depth} exceeds {TREE_MAX_DEPTH}")
        return t


def _dequantized(z):
    """Undoes octree/compression.py:88-136 on a loaded npz: `data` [n,2,2,2,3K+1] from `quant_colors` [K', 2^bits, 3] +
    `quant_map` [K', n,2,2,2] (+ `data_retained` [r, n,2,2,2,3] for the first r basis functions) + `sigma` [n,2,2,2]."""
    out = {k: z[k] for k in z.files}
    colors, qmap, sigma = z["quant_colors"].astype(np.float32), z["quant_map"].astype(np.int64), z["sigma"].astype(np.float32)
    retained = z["data_retained"].astype(np.float32) if "data_retained" in z.files else np.zeros((0,) + sigma.shape + (3,), np.float32)
    K = retained.shape[0] + colors.shape[0]
    rgb = np.empty(sigma.shape + (3, K), np.float32)                      # [..., channel, basis]: the layout of `data`
    for b in range(retained.shape[0]):
        rgb[..., b] = retained[b]
    for b in range(colors.shape[0]):
        rgb[..., retained.shape[0] + b] = colors[b][qmap[b]]
    out["data"] = np.concatenate([rgb.reshape(sigma.shape + (3 * K,)), sigma[..., None]], -1)
    return _NpzDict(out)


class _NpzDict(dict):
    """dict with the `.files` attribute of an NpzFile."""

    @property
    def files(self):
        return list(self.keys())


def parent_depth_from_child(child):
    """Rebuilds parent_depth [n,2] from the child offsets (compressed npz files drop it, compression.py:77)."""
    n = child.shape[0]
    pd = np.zeros((n, 2), np.int32)
    flat = child.reshape(n, 8)
    src, cell = np.nonzero(flat)
    dst = src + flat[src, cell]
    pd[dst, 0] = src * 8 + cell
    order = np.argsort(dst)                     # parents precede children (nodes are only ever appended)
    for s, d in zip(src[order], dst[order]):
        pd[d, 1] = pd[s, 1] + 1
    return pd


class _RenderPersp(torch.autograd.Function):
    """torch.autograd bridge so `mse.backward()` on a rendered image reaches tree.data as in
    octree/optimization.py:216-224; forward/backward are the HIP kernels."""

    @staticmethod
    def forward(ctx, data, renderer, c2w, width, height, fx, fy, opts):
        tree = renderer.tree
        ctx.args = (renderer, c2w, width, height, fx, fy, opts)
        out = oops.octree_render_persp(oops.tree_view(tree.child, data.detach(), tree.offset, tree.invradius), c2w, width,
                                       height, fx, opts, fy)
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        renderer, c2w, width, height, fx, fy, opts = ctx.args
        tree = renderer.tree
        grad = torch.zeros_like(tree.data.data)
        out, = ctx.saved_tensors
        oops.octree_render_persp_bwd(tree.view(), c2w, width, height, fx, opts, grad_out.contiguous(), grad, fy, out_rgb=out)
        return grad, None, None, None, None, None, None, None


class VolumeRenderer:
    """svox.VolumeRenderer for perspective cameras (no NDC: the LLFF forward-facing configs are out of scope)."""

    def __init__(self, tree, step_size=1e-3, background_brightness=1.0, ndc=None):
        if ndc is not None:
            raise NotImplementedError("NDC rendering (LLFF forward-facing scenes) is not supported")
        self.tree = tree
        self.step_size = float(step_size)
        self.background_brightness = float(background_brightness)

    def _opts(self, fast):
        thr = 1e-2 if fast else 0.0              # svox: fast = early stopping + skip of near-empty cells
        return oops.render_opts(self.step_size, self.background_brightness, thr, thr)

    def render_persp(self, c2w, width=800, height=800, fx=1111.111, fy=None, fast=False, cuda=True):
        """[H,W,3] image.  With grad mode on and a tree whose `data` requires grad the exact render (fast=False, as in
        octree/optimization.py:216) is differentiable; the early-stopping preset (fast=True, evaluation) never is."""
        c2w = torch.as_tensor(c2w, dtype=torch.float32, device=self.tree.device)
        data = self.tree.data
        if torch.is_grad_enabled() and data.requires_grad:
            if fast:
                # every fast=True call site of the reference sits under torch.no_grad() (octree/nerf/utils.py:276,
                # octree/evaluation.py:73); silently returning a detached image would surface later as an obscure
                # "does not require grad" error in the caller's backward
                raise oops.PxoError("render_persp(fast=True) is not differentiable: call it under torch.no_grad(), "
                                    "or use fast=False to optimise the tree")
            return _RenderPersp.apply(data, self, c2w, width, height, fx, fy, self._opts(False))
        return oops.octree_render_persp(self.tree.view(), c2w, width, height, fx, self._opts(fast), fy)

    def forward(self, origins, dirs, viewdirs, fast=False):
        """Colours [B,3] of explicit world-space rays (unit `dirs`)."""
        return oops.octree_render_rays(self.tree.view(), origins, dirs, viewdirs, self._opts(fast))

    __call__ = forward


# ---- svox.helpers._get_c_extension(): the slice of svox's native module the reference touches ---------------------------
class _RenderOptions:
    """svox.csrc RenderOptions as filled in by octree/extraction.py:184-195."""

    def __init__(self):
        self.step_size = 1e-3
        self.background_brightness = 1.0
        self.sigma_thresh = 0.0
        self.stop_thresh = 0.0
        self.ndc_width, self.ndc_height, self.ndc_focal = -1, -1, -1.0


class _CameraSpec:
    """svox.csrc CameraSpec (octree/extraction.py:197-203)."""

    def __init__(self):
        self.c2w = None
        self.fx = self.fy = 0.0
        self.width = self.height = 0


def _grid_weight_render(grid_data, cam, opts, offset, invradius):
    """_C.grid_weight_render (octree/extraction.py:205-211): (max compositing weight per voxel, hit mask) of one camera
    through the dense sigma grid [reso, reso, reso]."""
    if getattr(opts, "ndc_width", -1) > 0:
        raise NotImplementedError("NDC grid_weight_render (LLFF forward-facing scenes) is not supported")
    reso = grid_data.shape[0]
    o = oops.render_opts(opts.step_size, opts.background_brightness, opts.sigma_thresh, opts.stop_thresh)
    c2w = torch.as_tensor(cam.c2w, dtype=torch.float32, device=grid_data.device)[None, :3, :4]
    w = oops.grid_weight_render(grid_data.contiguous().reshape(-1), reso, c2w, cam.fx, cam.fy, cam.width, cam.height, o,
                                offset, invradius)
    w = w.reshape(reso, reso, reso)
    return w, w > 0


def _quantize_median_cut(data, weights, order):
    """_C.quantize_median_cut(data [n,3], weights [n] or empty, order) -> (colors [2^order, 3] float32, color_id_map [n]
    int32), as called by octree/compression.py:114-116.

    Median cut (Heckbert): `order` rounds, every box split at its (weighted) median along the axis of its largest extent,
    lower half first, so box b of a round becomes boxes 2b and 2b+1 of the next; a colour is the (weighted) mean of its
    box, a point's id the index of its box.  All boxes of a round are split at once (two stable sorts per round), in
    torch on whatever device `data` lives on.  svox's native implementation is not in the reference tree: this follows
    the published algorithm and the call site's contract (shapes, dtypes, id range), not svox's tie-breaking - parity
    unpinned, like the rest of the svox surface.  Boxes that run empty (n < 2^order) get colour 0."""
    x = torch.as_tensor(data).detach().to(torch.float32)
    if x.dim() != 2:
        raise ValueError("quantize_median_cut: data must be [n, channels]")
    n, dev = x.shape[0], x.device
    nbox = 1 << int(order)
    colors = torch.zeros(nbox, x.shape[1], dtype=torch.float32, device=dev)
    if n == 0:
        return colors, torch.zeros(0, dtype=torch.int32, device=dev)
    w = torch.as_tensor(weights).detach().to(torch.float32).to(dev).reshape(-1)
    weighted = w.numel() == n
    if not weighted:
        w = torch.ones(n, dtype=torch.float32, device=dev)
    box = torch.zeros(n, dtype=torch.int64, device=dev)
    for level in range(int(order)):
        nb = 1 << level
        lo = torch.full((nb, x.shape[1]), float("inf"), device=dev).scatter_reduce_(
            0, box[:, None].expand_as(x), x, "amin", include_self=True)
        hi = torch.full((nb, x.shape[1]), float("-inf"), device=dev).scatter_reduce_(
            0, box[:, None].expand_as(x), x, "amax", include_self=True)
        axis = (hi - lo).argmax(dim=1)                                   # per box; empty boxes give nan -> axis 0, unused
        key = x.gather(1, axis[box][:, None])[:, 0]
        order1 = torch.sort(key, stable=True).indices                    # by value ...
        order2 = torch.sort(box[order1], stable=True).indices            # ... then by box, values staying sorted
        perm = order1[order2]
        pbox = box[perm]
        start = torch.zeros(nb + 1, dtype=torch.int64, device=dev)
        start[1:] = torch.bincount(pbox, minlength=nb).cumsum(0)
        rank = torch.arange(n, device=dev) - start[pbox]
        size = (start[1:] - start[:-1])[pbox]
        if weighted:                                                      # first point at which half of the box's weight is reached
            cw = torch.cumsum(w[perm].double(), 0)
            base = torch.cat([torch.zeros(1, dtype=torch.float64, device=dev), cw])[start[pbox]]
            total = (torch.cat([torch.zeros(1, dtype=torch.float64, device=dev), cw])[start[1:]] - \
                     torch.cat([torch.zeros(1, dtype=torch.float64, device=dev), cw])[start[:-1]])[pbox]
            upper = (cw - base) > 0.5 * total
            upper &= rank > 0                                             # never leave the lower half empty ...
            upper |= (rank == size - 1) & (size > 1)                      # ... nor the upper one
        else:
            upper = rank >= size // 2                                     # nth_element at begin + size / 2
        new_box = torch.empty_like(box)
        new_box[perm] = 2 * pbox + upper.to(torch.int64)
        box = new_box
    wsum = torch.zeros(nbox, dtype=torch.float64, device=dev).index_add_(0, box, w.double())
    acc = torch.zeros(nbox, x.shape[1], dtype=torch.float64, device=dev).index_add_(0, box, x.double() * w.double()[:, None])
    filled = wsum > 0
    colors[filled] = (acc[filled] / wsum[filled][:, None]).float()
    return colors, box.to(torch.int32)


def _get_c_extension():
    return types.SimpleNamespace(RenderOptions=_RenderOptions, CameraSpec=_CameraSpec, grid_weight_render=_grid_weight_render,
                                 quantize_median_cut=_quantize_median_cut)


helpers = types.ModuleType(__name__ + ".helpers")
helpers._get_c_extension = _get_c_extension
helpers.__doc__ = "svox.helpers: `_get_c_extension()` as imported by octree/extraction.py:57 and octree/compression.py:34."


def install_as_svox():
    """Registers this module as `svox` (and `svox.helpers`), so that `import svox` / `from svox import N3Tree` /
    `from svox.helpers import _get_c_extension` in the reference's drivers resolve here."""
    import sys
    me = sys.modules[__name__]
    sys.modules["svox"] = me
    sys.modules["svox.helpers"] = helpers
    return me
