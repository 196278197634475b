"""CPU restatement of the PlenOctree side of the path: N3Tree build / sample / assign, the dense
grid weight render and the octree volume renderer (forward + gradient w.r.t. the tree data).

TEST INFRASTRUCTURE -- the checker, never the product.  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this.

PARITY UNPINNED.  These algorithms live in the third-party package `svox` (pin `svox>=0.2.28`,
/root/reference/environment.yml), which is NOT in /root/reference and not installed here.  What
is restated is svox's published algorithm (PlenOctrees, Yu et al. 2021, sec. 4.1-4.3 and the svox
0.2.28 semantics of N3Tree / VolumeRenderer / _C.grid_weight_render), anchored on the reference's
own call sites:

  N3Tree(N, data_dim, init_refine=0, depth_limit, radius, center, data_format)   octree/extraction.py:476-486
  tree[grid].refine()                                                            octree/extraction.py:341-350
  tree.depths / tree.max_depth / tree[inds].sample(S) / tree[inds] = rgba        octree/extraction.py:358-394
  tree[:, -1:].relu_(), tree.shrink_to_fit(), tree.save(path, compress=False)    octree/extraction.py:503-509
  _C.grid_weight_render(grid, cam, opts, offset, invradius)                      octree/extraction.py:181-214
  VolumeRenderer(t, step_size, ndc).render_persp(c2w, width, height, fx, fast)   octree/nerf/utils.py:456-474,
                                                                                 octree/optimization.py:174-216
  npz keys                                                                        octree/compression.py:76-86
It is pinned only by closed-form known answers (tests/test_octree_oracle.py).

Conventions (svox): world -> tree coordinates x_t = offset + invradius * x_w with
invradius = 0.5 / radius, offset = 0.5 * (1 - center / radius) (same formulas the reference uses at
octree/extraction.py:250-251); `child[n,i,j,k]` = index(child node) - n, 0 for a leaf;
`parent_depth[n]` = (packed index of the parent cell ((p*N+i)*N+j)*N+k, depth of n); leaves are ordered by
packed cell index.  All marching arithmetic is float32, one rounding per operation, in the order written
here (the HIP kernels are compiled without fused contraction so the two traversals take identical steps).
"""
import math

import numpy as np
import torch

from . import nerf_oracle as O

f32 = np.float32
N = 2


class RenderOptions:
    """svox RenderOptions as set by VolumeRenderer._get_options / extraction.calculate_grid_weights."""

    def __init__(self, step_size=1e-3, background_brightness=1.0, sigma_thresh=0.0, stop_thresh=0.0):
        self.step_size = f32(step_size)
        self.background_brightness = f32(background_brightness)
        self.sigma_thresh = f32(sigma_thresh)
        self.stop_thresh = f32(stop_thresh)

    @classmethod
    def for_renderer(cls, step_size, fast):
        """fast=True is svox's early-stopping preset (eval_octree passes fast=not no_early_stop)."""
        return cls(step_size, 1.0, 1e-2 if fast else 0.0, 1e-2 if fast else 0.0)


class Tree:
    def __init__(self, data_dim, depth_limit, center, radius):
        radius = np.broadcast_to(np.asarray(radius, f32), (3,)).astype(f32)
        center = np.broadcast_to(np.asarray(center, f32), (3,)).astype(f32)
        self.data_dim = data_dim
        self.depth_limit = depth_limit
        self.invradius = (f32(0.5) / radius).astype(f32)
        self.offset = (f32(0.5) * (f32(1.0) - center / radius)).astype(f32)
        self.child = np.zeros((1, N, N, N), np.int32)
        self.parent_depth = np.zeros((1, 2), np.int32)
        self.data = np.zeros((1, N, N, N, data_dim), f32)

    @property
    def n_internal(self):
        return self.child.shape[0]

    # -- structure ---------------------------------------------------------------------------
    def world2tree(self, p):
        return (self.offset + self.invradius * np.asarray(p, f32)).astype(f32)

    def query(self, p_tree):
        """svox query_single_from_root: (node, i, j, k, cube_sz, local position inside the leaf)."""
        x = np.clip(np.asarray(p_tree, f32), f32(0.0), f32(1.0 - 1e-6)).astype(f32)
        node, cube = 0, f32(N)
        while True:
            x = (x * f32(N)).astype(f32)
            u = np.floor(x).astype(np.int64)
            x = (x - u.astype(f32)).astype(f32)
            skip = int(self.child[node, u[0], u[1], u[2]])
            if skip == 0:
                return node, int(u[0]), int(u[1]), int(u[2]), cube, x
            cube = f32(cube * f32(N))
            node += skip

    def leaves(self):
        """All leaves in packed-index order: [n_leaves, 4] (node, i, j, k)."""
        return np.argwhere(self.child == 0)

    def depths(self):
        lv = self.leaves()
        return self.parent_depth[lv[:, 0], 1]

    def refine_leaves(self, packed):
        """N3Tree.refine(sel=...): split the given leaves (unique, ascending packed index)."""
        packed = np.unique(np.asarray(packed, np.int64))
        node, cell = packed // 8, packed % 8
        keep = self.parent_depth[node, 1] < self.depth_limit
        packed, node, cell = packed[keep], node[keep], cell[keep]
        n0, k = self.n_internal, packed.shape[0]
        if k == 0:
            return 0
        self.child = np.concatenate([self.child, np.zeros((k, N, N, N), np.int32)])
        self.data = np.concatenate([self.data, np.zeros((k, N, N, N, self.data_dim), f32)])
        self.parent_depth = np.concatenate([self.parent_depth, np.zeros((k, 2), np.int32)])
        new = np.arange(n0, n0 + k)
        self.child.reshape(-1)[packed] = (new - node).astype(np.int32)
        self.data[n0:] = self.data.reshape(-1, self.data_dim)[packed][:, None, None, None, :]
        self.parent_depth[n0:, 0] = packed
        self.parent_depth[n0:, 1] = self.parent_depth[node, 1] + 1
        return k

    def refine_at(self, pts_world):
        """tree[pts].refine() (octree/extraction.py:341-350)."""
        packed = []
        for p in np.asarray(pts_world, f32):
            n, i, j, k, _, _ = self.query(self.world2tree(p))
            packed.append(((n * 2 + i) * 2 + j) * 2 + k)
        return self.refine_leaves(packed)

    def node_corners(self):
        """Lower corner (tree coords, exact dyadic) and integer depth of every node."""
        corner = np.zeros((self.n_internal, 3), np.float64)
        for n in range(1, self.n_internal):
            packed, d = self.parent_depth[n]
            p, c = packed // 8, packed % 8
            ijk = np.array([(c >> 2) & 1, (c >> 1) & 1, c & 1], np.float64)
            corner[n] = corner[p] + ijk * 0.5 ** d
        return corner


def grid_points(reso, offset, invradius):
    """Cell centres of the reso^3 grid in world coordinates, x slowest (octree/extraction.py:294-303)."""
    arr = ((np.arange(reso, dtype=f32) + f32(0.5)) / f32(reso)).astype(f32)
    ax = [((arr - offset[a]) / invradius[a]).astype(f32) for a in range(3)]
    g = np.stack(np.meshgrid(*ax, indexing="ij"), -1)
    return g.reshape(-1, 3)


def build_from_mask(mask, depth, data_dim, center, radius):
    """Step 1's tree build, literally as the reference does it (octree/extraction.py:337-350):
    `depth` rounds of tree[grid].refine() with grid = centres of the masked voxels of the 2^(depth+1) grid."""
    reso = 2 ** (depth + 1)
    assert mask.shape == (reso, reso, reso)
    tree = Tree(data_dim, depth, center, radius)
    pts = grid_points(reso, tree.offset, tree.invradius)[mask.reshape(-1)]
    for _ in range(depth):
        tree.refine_at(pts)
    return tree


def leaf_corners(tree, leaves):
    """Lower corner and side length (tree coords) of leaves given as rows (node, i, j, k)."""
    corner = tree.node_corners()
    d = tree.parent_depth[leaves[:, 0], 1].astype(np.float64)
    side = 0.5 ** (d + 1)
    return corner[leaves[:, 0]] + leaves[:, 1:4] * side[:, None], side


# -- SH basis (svox calc_sh_basis == nerf_sh/nerf/sh.py:54-109 polynomials) ------------------------
def sh_basis_np(basis_dim, d):
    deg = int(round(math.sqrt(basis_dim))) - 1
    return O.sh_basis(deg, torch.tensor(np.asarray(d, f32))[None])[0].numpy().astype(f32)


# -- ray set-up --------------------------------------------------------------------------------
def cam2world_ray(ix, iy, c2w, W, H, fx, fy):
    """svox cam2world_ray: pixel centres at integer coordinates, -z forward (same convention as
    generate_rays, nerf_sh/nerf/utils.py:567-588); returns (origin, unit direction)."""
    c2w = np.asarray(c2w, f32)
    x = f32((f32(ix) - f32(0.5) * f32(W)) / f32(fx))
    y = f32(-(f32(iy) - f32(0.5) * f32(H)) / f32(fy))
    z = f32(np.sqrt(f32(f32(x * x) + f32(y * y)) + f32(1.0)))
    x = f32(x / z); y = f32(y / z); z = f32(f32(-1.0) / z)
    d = np.array([f32(f32(f32(c2w[a, 0] * x) + f32(c2w[a, 1] * y)) + f32(c2w[a, 2] * z)) for a in range(3)], f32)
    return c2w[:3, 3].astype(f32), d


def _dda_unit(cen, invdir):
    tmin, tmax = f32(0.0), f32(1e9)
    for a in range(3):
        t1 = f32(-cen[a] * invdir[a])
        t2 = f32(t1 + invdir[a])
        tmin = max(tmin, min(t1, t2))
        tmax = min(tmax, max(t1, t2))
    return f32(tmin), f32(tmax)


def _to_tree_ray(origin, direction, offset, invradius):
    o = np.array([f32(offset[a] + f32(invradius[a] * origin[a])) for a in range(3)], f32)
    d = np.array([f32(direction[a] * invradius[a]) for a in range(3)], f32)
    nrm = f32(np.sqrt(f32(f32(f32(d[0] * d[0]) + f32(d[1] * d[1])) + f32(d[2] * d[2]))))
    delta_scale = f32(f32(1.0) / nrm)
    d = (d * delta_scale).astype(f32)
    invdir = (f32(1.0) / (d + f32(1e-9)).astype(f32)).astype(f32)
    return o, d, invdir, delta_scale


def _sigmoid(x):
    return f32(1.0) / (f32(1.0) + np.exp(-x, dtype=f32))


def march_tree(tree, origin, direction, opt):
    """Sample sequence of svox trace_ray: list of (flat leaf index, dt_world) and the exit flag."""
    o, d, invdir, delta_scale = _to_tree_ray(origin, direction, tree.offset, tree.invradius)
    tmin, tmax = _dda_unit(o, invdir)
    if tmax < 0 or tmin > tmax:
        return None
    out, t = [], tmin
    while t < tmax:
        pos = np.array([f32(o[a] + f32(t * d[a])) for a in range(3)], f32)
        n, i, j, k, cube, local = tree.query(pos)
        s0, s1 = _dda_unit(local, invdir)
        delta_t = f32(f32(f32(s1 - s0) / cube) + opt.step_size)
        out.append((((n * 2 + i) * 2 + j) * 2 + k, f32(delta_t * delta_scale)))
        t = f32(t + delta_t)
    return out


def render_ray(tree, origin, direction, vdir, opt, want_samples=False):
    """svox trace_ray (SH format): returns rgb[3]."""
    basis_dim = (tree.data_dim - 1) // 3
    samples = march_tree(tree, origin, direction, opt)
    bg = opt.background_brightness
    if samples is None:
        return np.full(3, bg, f32)
    basis = sh_basis_np(basis_dim, vdir)
    flat = tree.data.reshape(-1, tree.data_dim)
    out, light = np.zeros(3, f32), f32(1.0)
    for leaf, dtw in samples:
        val = flat[leaf]
        sigma = val[-1]
        if sigma > opt.sigma_thresh:
            att = f32(np.exp(f32(-dtw * sigma), dtype=f32))
            weight = f32(light * f32(f32(1.0) - att))
            for c in range(3):
                tmp = f32(0.0)
                for q in range(basis_dim):
                    tmp = f32(tmp + f32(basis[q] * val[c * basis_dim + q]))
                out[c] = f32(out[c] + f32(weight * _sigmoid(tmp)))
            light = f32(light * att)
            if light <= opt.stop_thresh:
                scale = f32(f32(1.0) / f32(f32(1.0) - light))
                return (out * scale).astype(f32)
    return (out + f32(light * bg)).astype(f32)


def render_persp(tree, c2w, W, H, fx, opt, fy=None):
    fy = fx if fy is None else fy
    img = np.zeros((H, W, 3), f32)
    for iy in range(H):
        for ix in range(W):
            o, d = cam2world_ray(ix, iy, c2w, W, H, fx, fy)
            img[iy, ix] = render_ray(tree, o, d, d, opt)
    return img


def render_rays_torch(tree, data, origins, dirs, vdirs, opt, dtype=torch.float64):
    """Differentiable (w.r.t. `data`, a torch tensor shaped like tree.data) compositing over the sample
    sequence of march_tree, in `dtype`.  No early stop (training uses stop_thresh = 0)."""
    basis_dim = (tree.data_dim - 1) // 3
    flat = data.reshape(-1, tree.data_dim).to(dtype)
    outs = []
    for o, d, v in zip(np.asarray(origins, f32), np.asarray(dirs, f32), np.asarray(vdirs, f32)):
        samples = march_tree(tree, o, d, opt)
        bg = float(opt.background_brightness)
        if samples is None:
            outs.append(torch.full((3,), bg, dtype=dtype))
            continue
        idx = torch.tensor([s[0] for s in samples], dtype=torch.long)
        dtw = torch.tensor([float(s[1]) for s in samples], dtype=dtype)
        val = flat[idx]
        sigma = val[:, -1]
        live = sigma > float(opt.sigma_thresh)
        att = torch.where(live, torch.exp(-dtw * sigma), torch.ones_like(sigma))
        T = torch.cumprod(torch.cat([torch.ones(1, dtype=dtype), att]), 0)
        w = T[:-1] * (1.0 - att)
        basis = torch.tensor(sh_basis_np(basis_dim, v).astype(np.float64), dtype=dtype)
        rgb = torch.sigmoid((val[:, :-1].reshape(-1, 3, basis_dim) * basis).sum(-1))
        outs.append((w[:, None] * rgb).sum(0) + T[-1] * bg)
    return torch.stack(outs)


def grid_weight_render(sigma_grid, c2w, W, H, fx, opt, offset, invradius, fy=None, weight=None):
    """svox _C.grid_weight_render: per-voxel maximum compositing weight over the rays of one camera."""
    fy = fx if fy is None else fy
    reso = sigma_grid.shape[0]
    weight = np.zeros_like(sigma_grid, dtype=f32) if weight is None else weight
    cube = f32(reso)
    for iy in range(H):
        for ix in range(W):
            origin, direction = cam2world_ray(ix, iy, c2w, W, H, fx, fy)
            o, d, invdir, delta_scale = _to_tree_ray(origin, direction, offset, invradius)
            tmin, tmax = _dda_unit(o, invdir)
            if tmax < 0 or tmin > tmax:
                continue
            t, light = tmin, f32(1.0)
            while t < tmax:
                pos = np.array([f32(o[a] + f32(t * d[a])) for a in range(3)], f32)
                pos = np.clip(pos, f32(0.0), f32(1.0 - 1e-6)).astype(f32)
                pos = (pos * cube).astype(f32)
                u = np.floor(pos).astype(np.int64)
                local = (pos - u.astype(f32)).astype(f32)
                s0, s1 = _dda_unit(local, invdir)
                delta_t = f32(f32(f32(s1 - s0) / cube) + opt.step_size)
                sigma = sigma_grid[u[0], u[1], u[2]]
                if sigma > opt.sigma_thresh:
                    att = f32(np.exp(f32(-f32(delta_t * delta_scale) * sigma), dtype=f32))
                    w = f32(light * f32(f32(1.0) - att))
                    light = f32(light * att)
                    weight[u[0], u[1], u[2]] = max(weight[u[0], u[1], u[2]], w)
                    if light <= opt.stop_thresh:
                        break
                t = f32(t + delta_t)
    return weight
