"""Known-answer tests pinning oracle/octree_oracle.py (svox is third-party and absent: parity unpinned, so
the restatement is checked against closed forms and internal consistency) and CPU tests of the host-side
svox mirror (npz format, structure helpers)."""
import math
import os

import numpy as np
import pytest
import torch

from oracle import octree_oracle as T
from oracle import nerf_oracle as O

f32 = np.float32


def _pose(theta=30.0, phi=20.0, radius=4.0):
    from plenoctree_amd.nerf_sh.nerf.datasets import pose_spherical
    return pose_spherical(theta, phi, radius)


def _random_mask(depth, seed, p=0.08):
    reso = 2 ** (depth + 1)
    rs = np.random.RandomState(seed)
    return rs.rand(reso, reso, reso) < p


def test_single_voxel_mask_builds_a_chain():
    depth = 3
    reso = 2 ** (depth + 1)
    mask = np.zeros((reso,) * 3, bool)
    mask[5, 10, 3] = True
    t = T.build_from_mask(mask, depth, 4, [0, 0, 0], 1.5)
    assert t.n_internal == depth + 1                       # root + one node per level
    assert t.parent_depth[:, 1].tolist() == list(range(depth + 1))
    assert (t.child != 0).sum() == depth                   # one child pointer per internal level
    assert len(t.leaves()) == 7 * t.n_internal + 1
    # the masked voxel's centre lands in a depth-`depth` leaf whose cube is one voxel
    p = T.grid_points(reso, t.offset, t.invradius).reshape(reso, reso, reso, 3)[5, 10, 3]
    node, i, j, k, cube, local = t.query(t.world2tree(p))
    assert t.parent_depth[node, 1] == depth and cube == reso
    np.testing.assert_allclose(local, 0.5, atol=1e-4)
    # packed parent index decodes to the path of voxel (5,10,3): bits MSB first
    n = node
    for lvl in range(depth, 0, -1):
        packed = int(t.parent_depth[n, 0])
        cell = packed % 8
        # the cell of the depth-(lvl-1) node that holds node n is bit `depth+1-lvl` (from the LSB) of the voxel index
        shift = depth + 1 - lvl
        assert cell == (((5 >> shift) & 1) << 2 | ((10 >> shift) & 1) << 1 | ((3 >> shift) & 1))
        n = packed // 8
    assert n == 0


def test_tree_nodes_are_breadth_first_and_morton_sorted():
    depth = 3
    mask = _random_mask(depth, 1)
    t = T.build_from_mask(mask, depth, 4, [0.1, 0, -0.2], [1.4, 1.5, 1.3])
    d = t.parent_depth[:, 1]
    assert (np.diff(d) >= 0).all()
    for lvl in range(1, depth + 1):
        packed = t.parent_depth[d == lvl, 0]
        assert (np.diff(packed) > 0).all()                 # sorted by packed parent cell index
    # child offsets and parent_depth agree
    src, ci, cj, ck = np.nonzero(t.child)
    dst = src + t.child[src, ci, cj, ck]
    np.testing.assert_array_equal(t.parent_depth[dst, 0], ((src * 2 + ci) * 2 + cj) * 2 + ck)
    # node count per level = occupied cells of the mask pyramid
    reso = 2 ** (depth + 1)
    for lvl in range(1, depth + 1):
        f = reso // 2 ** lvl
        occ = mask.reshape(2 ** lvl, f, 2 ** lvl, f, 2 ** lvl, f).any(axis=(1, 3, 5))
        assert (d == lvl).sum() == occ.sum()


def test_leaf_corners_match_queries():
    depth = 2
    t = T.build_from_mask(_random_mask(depth, 3, 0.2), depth, 4, [0, 0, 0], 1.0)
    leaves = t.leaves()
    corner, side = T.leaf_corners(t, leaves)
    for (n, i, j, k), c, s in zip(leaves[::7], corner[::7], side[::7]):
        n2, i2, j2, k2, cube, _ = t.query((c + 0.25 * s).astype(f32))
        assert (n2, i2, j2, k2) == (n, i, j, k) and cube == round(1.0 / s)


def _uniform_tree(sigma, coeff0, basis_dim=1, radius=1.0):
    t = T.Tree(3 * basis_dim + 1, 2, [0, 0, 0], radius)
    t.data[..., -1] = sigma
    t.data[..., 0::basis_dim][..., :3] = 0.0
    for c in range(3):
        t.data[..., c * basis_dim] = coeff0[c]
    return t


def test_render_uniform_cube_closed_form():
    """Homogeneous medium in the root node: T = exp(-sigma * sum(dt_world)).  Every sample starts `step` past the
    previous cell's exit, so the steps telescope: sum(dt_tree) = chord + step."""
    sigma, step = 0.7, 1e-3
    c0 = np.array([0.3, -0.2, 1.0], f32)
    t = _uniform_tree(sigma, c0, basis_dim=1, radius=1.0)
    opt = T.RenderOptions(step_size=step, background_brightness=1.0)
    origin = np.array([0.0, -3.0, 0.1], f32)
    d = np.array([0.05, 1.0, 0.02], f32); d /= np.linalg.norm(d)
    rgb = T.render_ray(t, origin, d, d, opt)
    samples = T.march_tree(t, origin, d, opt)
    assert len(samples) == 2                                # two root cells along y
    total = sum(float(s[1]) for s in samples)
    chord_world = 2.0 / abs(d[1]) if abs(d[1]) > max(abs(d[0]), abs(d[2])) else None
    assert abs(total - (chord_world + step / 0.5)) < 1e-5              # invradius 0.5: dt_world = dt_tree / 0.5
    trans = math.exp(-sigma * total)
    col = 1.0 / (1.0 + np.exp(-0.28209479177387814 * c0))
    np.testing.assert_allclose(rgb, (1 - trans) * col + trans * 1.0, rtol=2e-6, atol=2e-6)


def test_render_miss_and_empty_tree_return_background():
    t = _uniform_tree(0.0, [0, 0, 0])
    opt = T.RenderOptions(1e-3, background_brightness=0.25)
    d = np.array([0.0, 0.0, 1.0], f32)
    np.testing.assert_array_equal(T.render_ray(t, np.array([5, 5, -4], f32), d, d, opt), np.full(3, 0.25, f32))     # misses
    np.testing.assert_allclose(T.render_ray(t, np.array([0.1, 0.2, -4], f32), d, d, opt), 0.25, atol=1e-7)        # sigma = 0


def test_render_opaque_leaf_and_early_stop():
    t = _uniform_tree(1e4, [2.0, 0.0, -2.0])
    d = np.array([0.0, 0.0, 1.0], f32)
    o = np.array([0.3, 0.3, -4.0], f32)
    col = 1.0 / (1.0 + np.exp(-0.28209479177387814 * np.array([2.0, 0.0, -2.0])))
    exact = T.render_ray(t, o, d, d, T.RenderOptions(1e-3))
    fast = T.render_ray(t, o, d, d, T.RenderOptions.for_renderer(1e-3, fast=True))
    np.testing.assert_allclose(exact, col, atol=1e-6)
    np.testing.assert_allclose(fast, col, atol=1e-6)       # early stop rescales by 1/(1-T)


def test_sh_view_dependence_matches_eval_sh():
    rs = np.random.RandomState(0)
    for K in (4, 9, 16, 25):
        t = T.Tree(3 * K + 1, 2, [0, 0, 0], 1.0)
        coeff = rs.randn(3 * K).astype(f32)
        t.data[..., :-1] = coeff
        t.data[..., -1] = 1e4                              # opaque: colour = sigmoid(eval_sh)
        d = rs.randn(3).astype(f32); d /= np.linalg.norm(d)
        o = (-3.0 * d).astype(f32)
        rgb = T.render_ray(t, o, d, d, T.RenderOptions(1e-3))
        deg = int(math.isqrt(K)) - 1
        ref = torch.sigmoid(O.eval_sh(deg, torch.tensor(coeff).reshape(3, K), torch.tensor(d))).numpy()
        np.testing.assert_allclose(rgb, ref, atol=2e-6)


def test_render_gradient_matches_finite_differences():
    depth = 2
    t = T.build_from_mask(_random_mask(depth, 5, 0.3), depth, 13, [0, 0, 0], 1.2)
    rs = np.random.RandomState(2)
    t.data[:] = (rs.randn(*t.data.shape) * 0.5).astype(f32)
    t.data[..., -1] = np.abs(t.data[..., -1]) * 4 + 0.1
    opt = T.RenderOptions(1e-3)
    rays_o = np.array([[0.2, -3.0, 0.1], [2.5, 2.0, 1.0]], f32)
    rays_d = np.array([[0.0, 1.0, 0.05], [-0.7, -0.6, -0.3]], f32)
    rays_d /= np.linalg.norm(rays_d, axis=1, keepdims=True)
    g = torch.tensor(rs.randn(2, 3))
    data = torch.tensor(t.data.astype(np.float64), requires_grad=True)
    out = T.render_rays_torch(t, data, rays_o, rays_d, rays_d, opt)
    (out * g).sum().backward()
    grad = data.grad.reshape(-1)
    # f32 composite agrees with the f64 differentiable one
    f32_out = np.stack([T.render_ray(t, o, d, d, opt) for o, d in zip(rays_o, rays_d)])
    np.testing.assert_allclose(f32_out, out.detach().numpy(), atol=5e-6)
    idx = torch.nonzero(grad.abs() > 1e-6).reshape(-1)[::17][:12]
    assert len(idx) >= 6
    flat0 = data.detach().reshape(-1)
    for i in idx.tolist():
        for eps in (1e-4,):
            p, m = flat0.clone(), flat0.clone()
            p[i] += eps; m[i] -= eps
            fp = (T.render_rays_torch(t, p.reshape(data.shape), rays_o, rays_d, rays_d, opt) * g).sum()
            fm = (T.render_rays_torch(t, m.reshape(data.shape), rays_o, rays_d, rays_d, opt) * g).sum()
            fd = float(fp - fm) / (2 * eps)
            assert abs(fd - float(grad[i])) <= 1e-5 + 1e-4 * abs(fd), (i, fd, float(grad[i]))


def test_grid_weight_render_known_answers():
    reso = 8
    offset, invradius = np.full(3, 0.5, f32), np.full(3, 0.5, f32)       # centre 0, radius 1
    c2w = _pose(0.0, 0.0, 4.0)
    opt = T.RenderOptions(1e-3)
    # empty grid: no weight anywhere
    w = T.grid_weight_render(np.zeros((reso,) * 3, f32), c2w, 6, 6, 8.0, opt, offset, invradius)
    assert w.max() == 0.0
    # opaque grid: only voxels on the faces seen by the camera get weight ~1, interior stays 0
    w = T.grid_weight_render(np.full((reso,) * 3, 1e4, f32), c2w, 24, 24, 30.0, opt, offset, invradius)
    assert w.max() > 0.999
    assert w[1:-1, 1:-1, 1:-1].max() < 1e-6
    # homogeneous thin medium: the first voxel a ray enters has weight 1 - exp(-sigma * dt)
    sigma = 0.05
    w = T.grid_weight_render(np.full((reso,) * 3, sigma, f32), c2w, 16, 16, 20.0, opt, offset, invradius)
    assert 0 < w.max() <= 1 - math.exp(-sigma * (math.sqrt(3) * 2.0 / reso + 2e-3 / 0.5)) + 1e-6


# ---- host-side svox mirror (CPU) ------------------------------------------------------------------------
def _host_tree_from_oracle(t, fmt):
    from plenoctree_amd.octree.svox import N3Tree
    radius = 0.5 / t.invradius
    center = (1.0 - 2.0 * t.offset) * radius
    h = N3Tree(N=2, data_dim=t.data_dim, depth_limit=t.depth_limit, radius=radius, center=center, data_format=fmt)
    h.child = torch.from_numpy(t.child.copy())
    h.parent_depth = torch.from_numpy(t.parent_depth.copy())
    h.data = torch.from_numpy(t.data.copy())
    h.level_nodes = np.bincount(t.parent_depth[:, 1]).tolist()
    return h


def test_npz_format_round_trip(tmp_path):
    from plenoctree_amd.octree import svox
    depth = 3
    t = T.build_from_mask(_random_mask(depth, 9), depth, 49, [0.1, 0.2, -0.1], [1.0, 1.2, 0.9])
    t.data[:] = np.random.RandomState(0).randn(*t.data.shape).astype(f32)
    h = _host_tree_from_oracle(t, "SH16")
    np.testing.assert_allclose(h.offset, t.offset, rtol=1e-6)
    np.testing.assert_allclose(h.invradius, t.invradius, rtol=1e-6)
    assert h.n_leaves == len(t.leaves()) and h.max_depth == depth
    path = os.path.join(str(tmp_path), "tree.npz")
    h.save(path, compress=False)
    z = np.load(path)
    # exactly the keys svox writes (octree/compression.py:76-86 deletes five of them and keeps the rest)
    assert sorted(z.files) == sorted(["data_dim", "child", "parent_depth", "n_internal", "n_free", "invradius3", "offset",
                                      "depth_limit", "geom_resize_fact", "data", "data_format"])
    assert z["data"].dtype == np.float16 and z["child"].dtype == np.int32 and z["parent_depth"].dtype == np.int32
    assert z["child"].shape == (t.n_internal, 2, 2, 2) and int(z["n_internal"]) == t.n_internal and int(z["n_free"]) == 0
    assert str(z["data_format"]) == "SH16" and int(z["data_dim"]) == 49
    back = svox.N3Tree.load(path)
    assert torch.equal(back.child, h.child) and torch.equal(back.parent_depth, h.parent_depth)
    assert torch.equal(back.data, h.data.half().float())
    assert back.level_nodes == h.level_nodes and back.depth_limit == depth and back.data_format == "SH16"
    assert "capacity:%d/%d" % (t.n_internal, t.n_internal) in repr(back).split()[3]      # octree/task_manager.py:108-109
    # a file without parent_depth (compression.py:77 drops it) is still loadable
    zz = dict(z); del zz["parent_depth"], zz["n_internal"], zz["n_free"], zz["depth_limit"], zz["geom_resize_fact"]
    np.savez(path, **zz)
    again = svox.N3Tree.load(path)
    assert torch.equal(again.parent_depth, h.parent_depth)


def test_host_mirror_rejects_unsupported_configurations():
    from plenoctree_amd.octree import svox
    with pytest.raises(NotImplementedError):
        svox.N3Tree(N=4, data_dim=49, data_format="SH16")
    with pytest.raises(NotImplementedError):
        svox.N3Tree(N=2, data_dim=4, data_format="RGBA")
    with pytest.raises(NotImplementedError):
        svox.N3Tree(N=2, data_dim=76, data_format="SG25")
    with pytest.raises(ValueError):
        svox.N3Tree(N=2, data_dim=50, data_format="SH16")
    with pytest.raises(ValueError):
        svox.N3Tree(N=2, data_dim=49, data_format="SH16", depth_limit=11)
    t = svox.N3Tree(N=2, data_dim=49, data_format="SH16", depth_limit=8, radius=[1.0, 2.0, 4.0], center=[0.5, 0, -1])
    np.testing.assert_allclose(t.invradius, [0.5, 0.25, 0.125])
    np.testing.assert_allclose(t.offset, [0.25, 0.5, 0.625])
    with pytest.raises(NotImplementedError):
        svox.VolumeRenderer(t, ndc=object())
