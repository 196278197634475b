"""Host-side mirror of the slice of `svox` the reference uses (N3Tree, VolumeRenderer) on the MI355X path.

svox (pin svox>=0.2.28) is a third-party dependency of the reference and is not part of its tree; the
classes below keep its names, argument meaning and npz format for exactly the calls the reference makes:

    N3Tree(N, data_dim, init_refine=0, depth_limit, radius, center, data_format, map_location)   octree/extraction.py:476-486
    tree[grid].refine() x depth          -> N3Tree.refine_from_mask(mask)                        octree/extraction.py:337-352
    tree.depths == tree.max_depth, tree[inds].sample(S), tree[inds] = rgba                        octree/extraction.py:358-394
                                         -> max_depth_nodes(), sample_max_depth_cells(), max_depth_data()
    tree[:, -1:].relu_(), shrink_to_fit(), save(path, compress=False), N3Tree.load(path)          octree/extraction.py:503-509,
                                                                                                 octree/evaluation.py:81
    VolumeRenderer(t, step_size, ndc=None).render_persp(c2w, height, width, fx, fast, cuda)       octree/nerf/utils.py:456-474,
                                                                                                 octree/optimization.py:174-216
Storage and node order are svox's (include/plenoctree_octree.h).  All arithmetic runs in libplenoctree_hip.so
through plenoctree_amd.octree_ops; torch holds the device arrays.  Only N = 2 and the SH data formats
(`SH1/4/9/16/25`) are supported -- anything else raises, there is no fallback.
"""
import re

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


def parse_data_format(fmt, data_dim):
    """'SH16' -> basis_dim 16 (svox DataFormat)."""
    m = re.fullmatch(r"SH(\d+)", str(fmt))
    if not m:
        raise NotImplementedError(f"data_format {fmt!r}: only the spherical-harmonics formats SH1/4/9/16/25 are supported")
    k = int(m.group(1))
    if k not in (1, 4, 9, 16, 25):
        raise ValueError(f"data_format {fmt}: basis_dim must be a square <= 25")
    if data_dim is not None and data_dim != 3 * k + 1:
        raise ValueError(f"data_dim {data_dim} does not match data_format {fmt} (expected {3 * k + 1})")
    return k


class N3Tree:
    def __init__(self, N=2, data_dim=None, depth_limit=10, init_reserve=1, init_refine=0, geom_resize_fact=1.0,
                 radius=0.5, center=(0.5, 0.5, 0.5), data_format="RGBA", extra_data=None, device="cpu",
                 map_location=None):
        if N != 2:
            raise NotImplementedError("only octrees (tree_branch_n = 2) are supported")
        if init_refine != 0:
            raise NotImplementedError("init_refine must be 0 (octree/extraction.py:478)")
        if extra_data is not None:
            raise NotImplementedError("extra_data (spherical-gaussian formats) is not supported")
        if not 1 <= depth_limit <= TREE_MAX_DEPTH:
            raise ValueError(f"depth_limit must be in [1, {TREE_MAX_DEPTH}]")
        self.N = 2
        self.basis_dim = parse_data_format(data_format, data_dim)
        self.data_dim = 3 * self.basis_dim + 1
        self.data_format = f"SH{self.basis_dim}"
        self.depth_limit = int(depth_limit)
        self.geom_resize_fact = float(geom_resize_fact)
        dev = torch.device(map_location if map_location is not None else device)
        radius, center = _vec3(radius, "radius"), _vec3(center, "center")
        self.invradius = (np.float32(0.5) / radius).astype(np.float32)
        self.offset = (np.float32(0.5) * (np.float32(1.0) - center / radius)).astype(np.float32)
        self.child = torch.zeros(1, 2, 2, 2, dtype=torch.int32, device=dev)
        self.parent_depth = torch.zeros(1, 2, dtype=torch.int32, device=dev)
        self.data = torch.zeros(1, 2, 2, 2, self.data_dim, dtype=torch.float32, device=dev)
        self.level_nodes = [1]

    # ---- structure -------------------------------------------------------------------------------
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

    def __repr__(self):
        return (f"svox.N3Tree(N={self.N}, data_dim={self.data_dim}, depth_limit={self.depth_limit}, "
                f"capacity:{self.n_internal}/{self.capacity}, data_format:{self.data_format})")

    def refine_from_mask(self, mask):
        """Replaces `for _ in range(depth): tree[grid].refine()` (octree/extraction.py:337-350), where grid are
        the centres of the masked voxels of the 2^(depth_limit+1) grid: `mask` is that grid's uint8 mask
        (x slowest).  Must be called on a fresh tree (root only)."""
        if self.n_internal != 1:
            raise PxoError("refine_from_mask needs a fresh tree")
        self.child, self.parent_depth, self.level_nodes = oops.tree_from_mask(mask.reshape(-1), self.depth_limit)
        while len(self.level_nodes) > 1 and self.level_nodes[-1] == 0:
            self.level_nodes.pop()
        self.data = torch.zeros(self.n_internal, 2, 2, 2, self.data_dim, dtype=torch.float32, device=mask.device)
        return self

    def max_depth_nodes(self):
        """(first node, count) of the deepest level: its 8*count cells are the leaves with depth == max_depth,
        in leaf order (`tree.depths == tree.max_depth`, octree/extraction.py:358-360)."""
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
        return self.data[node0 + first: node0 + first + count].view(count * 8, self.data_dim)

    def relu_sigma_(self):
        """tree[:, -1:].relu_() (octree/extraction.py:503)."""
        oops.tree_relu_sigma(self.data)
        return self

    def shrink_to_fit(self):
        return self                              # arrays are always exactly n_internal long

    def view(self):
        return oops.tree_view(self.child, self.data, self.offset, self.invradius)

    def clone(self, device=None):
        t = object.__new__(N3Tree)
        t.__dict__.update(self.__dict__)
        dev = torch.device(device) if device is not None else self.device
        t.child, t.parent_depth, t.data = (x.detach().clone().to(dev) for x in (self.child, self.parent_depth, self.data))
        t.level_nodes = list(self.level_nodes)
        return t

    # ---- npz (svox N3Tree.save / load; keys as consumed by octree/compression.py:76-86) ------------------
    def save(self, path, shrink=True, compress=True):
        z = {
            "data_dim": self.data_dim,
            "child": self.child.cpu().numpy(),
            "parent_depth": self.parent_depth.cpu().numpy(),
            "n_internal": self.n_internal,
            "n_free": 0,
            "invradius3": self.invradius,
            "offset": self.offset,
            "depth_limit": self.depth_limit,
            "geom_resize_fact": self.geom_resize_fact,
            "data": self.data.detach().half().cpu().numpy(),       # svox stores float16
            "data_format": self.data_format,
        }
        (np.savez_compressed if compress else np.savez)(path, **z)

    @classmethod
    def load(cls, path, device="cpu", map_location=None):
        dev = torch.device(map_location if map_location is not None else device)
        z = np.load(path)
        if "quant_colors" in z.files:
            raise NotImplementedError("median-cut compressed trees (octree/compression.py) are not supported")
        t = object.__new__(cls)
        t.N = 2
        t.data_dim = int(z["data_dim"])
        fmt = str(z["data_format"]) if "data_format" in z.files else "RGBA"
        t.basis_dim = parse_data_format(fmt, t.data_dim)
        t.data_format = f"SH{t.basis_dim}"
        child = z["child"]
        if child.shape[1:] != (2, 2, 2):
            raise NotImplementedError("only octrees (N = 2) are supported")
        n = int(z["n_internal"]) if "n_internal" in z.files else child.shape[0]
        t.child = torch.from_numpy(child[:n].astype(np.int32)).to(dev)
        t.data = torch.from_numpy(z["data"][:n].astype(np.float32)).to(dev)
        if "parent_depth" in z.files:
            pd = z["parent_depth"][:n].astype(np.int32)
        else:
            pd = parent_depth_from_child(child[:n])
        t.parent_depth = torch.from_numpy(pd).to(dev)
        if "invradius3" in z.files:
            t.invradius = z["invradius3"].astype(np.float32)
        else:
            t.invradius = np.repeat(np.float32(z["invradius"]), 3)
        t.offset = z["offset"].astype(np.float32)
        t.depth_limit = int(z["depth_limit"]) if "depth_limit" in z.files else int(pd[:, 1].max())
        t.geom_resize_fact = float(z["geom_resize_fact"]) if "geom_resize_fact" in z.files else 1.0
        counts = np.bincount(pd[:, 1])
        if (np.diff(pd[:, 1]) < 0).any():
            raise NotImplementedError("nodes must be stored breadth-first (true for every tree built level by level)")
        t.level_nodes = [int(c) for c in counts]
        if t.max_depth > TREE_MAX_DEPTH:
            raise NotImplementedError(f"tree depth {t.max_depth} exceeds {TREE_MAX_DEPTH}")
        return t


def parent_depth_from_child(child):
    """Rebuilds parent_depth [n,2] from the child offsets (compressed npz files drop it, compression.py:77)."""
    n = child.shape[0]
    pd = np.zeros((n, 2), np.int32)
    flat = child.reshape(n, 8)
    src, cell = np.nonzero(flat)
    dst = src + flat[src, cell]
    pd[dst, 0] = src * 8 + cell
    order = np.argsort(dst)                     # parents precede children in breadth-first storage
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
        out = oops.octree_render_persp(oops.tree_view(tree.child, data, tree.offset, tree.invradius), c2w, width, height, fx,
                                       opts, fy)
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        renderer, c2w, width, height, fx, fy, opts = ctx.args
        tree = renderer.tree
        grad = torch.zeros_like(tree.data)
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
        """[H,W,3] image.  With a tree whose `data` requires grad the result is differentiable
        (exact marching; `fast` must be False, as in octree/optimization.py:216)."""
        c2w = torch.as_tensor(c2w, dtype=torch.float32, device=self.tree.device)
        data = self.tree.data
        if torch.is_grad_enabled() and data.requires_grad:
            if fast:
                raise PxoError("the gradient is defined for exact marching only (fast=False)")
            return _RenderPersp.apply(data, self, c2w, width, height, fx, fy, self._opts(False))
        return oops.octree_render_persp(self.tree.view(), c2w, width, height, fx, self._opts(fast), fy)

    def forward(self, origins, dirs, viewdirs, fast=False):
        """Colours [B,3] of explicit world-space rays (unit `dirs`)."""
        return oops.octree_render_rays(self.tree.view(), origins, dirs, viewdirs, self._opts(fast))

    __call__ = forward
