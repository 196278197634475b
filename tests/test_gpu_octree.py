"""GPU parity tests of the PlenOctree side (include/plenoctree_octree.h) against oracle/octree_oracle.py.

Bars: tree structure (child, parent_depth) bit-exact; sample points, weights and rendered colours within
float32 round-off of the oracle (atol 2e-5: the marching takes identical steps, only exp/sigmoid and the
SH summation order differ); gradients rtol 2e-3 against the float64 autograd oracle (float32 atomics)."""
import math
import os

import numpy as np
import pytest
import torch

from oracle import nerf_oracle as O
from oracle import octree_oracle as T
from _helpers import _gpu, close, make_params

pytestmark = pytest.mark.gpu
f32 = np.float32


def _oops():
    from plenoctree_amd import octree_ops
    return octree_ops


def _pose(theta, phi, radius=4.0):
    from plenoctree_amd.nerf_sh.nerf.datasets import pose_spherical
    return pose_spherical(theta, phi, radius)


def _mask(depth, seed, p):
    reso = 2 ** (depth + 1)
    return np.random.RandomState(seed).rand(reso, reso, reso) < p


def _random_tree(depth, seed, K, p=0.15, center=(0.1, 0.0, -0.2), radius=(1.4, 1.5, 1.3)):
    """Oracle tree with random SH data; about a third of the leaves are empty (sigma <= 0)."""
    t = T.build_from_mask(_mask(depth, seed, p), depth, 3 * K + 1, center, radius)
    rs = np.random.RandomState(seed + 100)
    t.data[:] = (rs.randn(*t.data.shape) * 0.7).astype(f32)
    t.data[..., -1] = ((rs.rand(*t.data.shape[:-1]) - 0.35) * 12.0).astype(f32)
    return t


def _device_tree(t, dev):
    oops = _oops()
    child = torch.from_numpy(t.child).to(dev)
    data = torch.from_numpy(t.data).to(dev)
    return oops.tree_view(child, data, t.offset, t.invradius), (child, data)


# ---------------------------------------------------------------------------------------
def test_tree_build_matches_oracle_bit_exact():
    oops = _oops(); dev = _gpu()
    cases = [(1, _mask(1, 0, 0.5)), (2, _mask(2, 1, 0.2)), (3, _mask(3, 2, 0.08)), (3, _mask(3, 3, 0.9)),
             (3, np.ones((16,) * 3, bool)), (2, np.zeros((8,) * 3, bool))]
    single = np.zeros((16,) * 3, bool); single[5, 10, 3] = True
    cases.append((3, single))
    for depth, mask in cases:
        ref = T.build_from_mask(mask, depth, 4, [0, 0, 0], 1.5)
        child, pd, levels = oops.tree_from_mask(torch.from_numpy(mask.astype(np.uint8)).to(dev), depth)
        want_levels = np.bincount(ref.parent_depth[:, 1], minlength=depth + 1).tolist()
        assert levels == want_levels, (levels, want_levels)
        assert child.shape == (ref.n_internal, 2, 2, 2) and child.dtype == torch.int32
        assert np.array_equal(child.cpu().numpy(), ref.child)
        assert np.array_equal(pd.cpu().numpy(), ref.parent_depth)


def test_tree_build_full_size_properties():
    """512^3 mask (init_grid_depth 8, the reference default): level counts equal the occupancy pyramid,
    child/parent links are mutually consistent, nodes are breadth-first and Morton-sorted."""
    oops = _oops(); dev = _gpu()
    depth, reso = 8, 512
    g = torch.Generator(device="cpu").manual_seed(0)
    # a thick spherical shell + sparse noise: ~2 % of the voxels
    ax = (torch.arange(reso, dtype=torch.float32) + 0.5) / reso - 0.5
    r = (ax[:, None, None] ** 2 + ax[None, :, None] ** 2 + ax[None, None, :] ** 2).sqrt()
    mask = ((r > 0.30) & (r < 0.32)) | (torch.rand(reso, reso, reso, generator=g) < 2e-4)
    mask_d = mask.to(torch.uint8).to(dev)
    torch.cuda.synchronize()
    child, pd, levels = oops.tree_from_mask(mask_d, depth)
    torch.cuda.synchronize()
    occ = mask_d.bool()
    for lvl in range(depth, 0, -1):
        occ = occ.reshape(2 ** lvl, 2, 2 ** lvl, 2, 2 ** lvl, 2).any(dim=5).any(dim=3).any(dim=1)
        assert levels[lvl] == int(occ.sum()), lvl
    n = sum(levels)
    assert child.shape[0] == n and levels[0] == 1
    d = pd[:, 1].long()
    assert bool((d[1:] >= d[:-1]).all())
    flat = child.reshape(n, 8).long()
    src, cell = torch.nonzero(flat, as_tuple=True)
    dst = src + flat[src, cell]
    assert dst.numel() == n - 1 and bool((dst[1:] > dst[:-1]).all())          # every non-root node has one parent
    assert torch.equal(pd[dst, 0].long(), src * 8 + cell)
    assert torch.equal(d[dst], d[src] + 1)
    assert not bool(flat[d == depth].any())                                  # deepest level: all leaves


def test_sample_cells_and_relu_and_threshold():
    oops = _oops(); dev = _gpu()
    depth = 3
    t = T.build_from_mask(_mask(depth, 4, 0.1), depth, 4, [0.1, 0.0, -0.2], [1.4, 1.5, 1.3])
    pd = torch.from_numpy(t.parent_depth).to(dev)
    levels = np.bincount(t.parent_depth[:, 1])
    node0, n_nodes, S = int(t.n_internal - levels[-1]), int(levels[-1]), 5
    u = torch.rand(n_nodes * 8 * S, 3, generator=torch.Generator().manual_seed(1))
    pts = oops.tree_sample_cells(pd, node0, n_nodes, S, t.offset, t.invradius, u=u.to(dev))
    assert pts.shape == (n_nodes * 8, S, 3)
    leaves = np.array([[n, (c >> 2) & 1, (c >> 1) & 1, c & 1] for n in range(node0, t.n_internal) for c in range(8)])
    corner, side = T.leaf_corners(t, leaves)
    un = u.numpy().reshape(n_nodes * 8, S, 3)
    want = ((corner[:, None, :].astype(f32) + un * side[:, None, None].astype(f32)) - t.offset) / t.invradius
    close("sample points", pts, torch.from_numpy(want.astype(f32)), rtol=1e-6, atol=1e-6)
    # every point falls into its own cell
    for q in range(0, n_nodes * 8, 37):
        n, i, j, k, cube, _ = t.query(t.world2tree(pts[q, S // 2].cpu().numpy()))
        assert [n, i, j, k] == leaves[q].tolist()
    # Philox path: deterministic per seed, uniform inside the cells
    a = oops.tree_sample_cells(pd, node0, n_nodes, S, t.offset, t.invradius, seed=3)
    b = oops.tree_sample_cells(pd, node0, n_nodes, S, t.offset, t.invradius, seed=3)
    c = oops.tree_sample_cells(pd, node0, n_nodes, S, t.offset, t.invradius, seed=4)
    assert torch.equal(a, b) and not torch.equal(a, c)
    data = torch.randn(6, 2, 2, 2, 13, device=dev)
    ref = data.clone(); ref[..., -1].clamp_(min=0)
    oops.tree_relu_sigma(data)
    assert torch.equal(data, ref)
    v = torch.randn(1001, device=dev)
    assert torch.equal(oops.threshold_mask(v, 0.25), (v >= 0.25).to(torch.uint8))


@pytest.mark.parametrize("reso,W,H,fx", [(16, 13, 11, 14.0), (64, 13, 11, 14.0), (64, 40, 36, 70.0), (12, 13, 11, 14.0)])
def test_grid_weight_render_matches_oracle(reso, W, H, fx):
    """reso 16: the whole grid sits in the slab kernel's LDS window; reso 64 at 13x11 pixels: a tile spans the grid, so most
    samples take the kernel's out-of-window path; 64 at 40x36: several tiles, samples mostly inside the window; 12: not a power
    of two, the plain bricked kernel."""
    oops = _oops(); dev = _gpu()
    rs = np.random.RandomState(0)
    sigma = ((rs.rand(reso, reso, reso) - 0.6) * 30.0).astype(f32)
    t = T.Tree(4, 3, [0.1, 0.0, -0.2], [1.4, 1.5, 1.3])
    cams = np.stack([_pose(20.0, 30.0), _pose(200.0, -10.0), _pose(100.0, 60.0)])
    opt = T.RenderOptions(step_size=1e-3)
    want = np.zeros_like(sigma)
    for c in cams:
        T.grid_weight_render(sigma, c, W, H, fx, opt, t.offset, t.invradius, weight=want)
    got = oops.grid_weight_render(torch.from_numpy(sigma).to(dev), reso, torch.from_numpy(cams).to(dev), fx, fx, W, H,
                                  oops.render_opts(1e-3), t.offset, t.invradius)
    assert (want > 0).sum() > 50
    close("grid weights", got, torch.from_numpy(want), rtol=1e-5, atol=1e-6)
    # the slab-staged kernel and the per-sample kernel take the same samples with the same arithmetic: bit-equal (the default
    # call above picked one of them on the device by the fraction of voxels above sigma_thresh; the tuning knob forces either)
    forced = {}
    try:
        for mode in ("0", "1"):
            oops.set_tuning(oops.TUNE_GW_MARCHER, int(mode))
            forced[mode] = oops.grid_weight_render(torch.from_numpy(sigma).to(dev), reso, torch.from_numpy(cams).to(dev), fx, fx,
                                                   W, H, oops.render_opts(1e-3), t.offset, t.invradius)
    finally:
        oops.set_tuning(oops.TUNE_GW_MARCHER, -1)
    with pytest.raises(Exception):
        oops.set_tuning(oops.TUNE_BWD_CACHE_ROWS, 12)            # not an instantiation: rejected, not silently remapped
    assert torch.equal(forced["0"], forced["1"]) and torch.equal(forced["0"], got)
    # ... and which workgroup takes which tile does not matter either (PXO_TUNE_GW_TILE_ORDER: XCD supertiles, padded grid)
    if reso % 4 == 0 and (reso & (reso - 1)) == 0:
        default_order = oops.get_tuning(oops.TUNE_GW_TILE_ORDER)
        try:
            for order in (0, 1):
                for mode in (0, 1):
                    oops.set_tuning(oops.TUNE_GW_TILE_ORDER, order); oops.set_tuning(oops.TUNE_GW_MARCHER, mode)
                    w = oops.grid_weight_render(torch.from_numpy(sigma).to(dev), reso, torch.from_numpy(cams).to(dev), fx, fx, W, H,
                                                oops.render_opts(1e-3), t.offset, t.invradius)
                    assert torch.equal(w, got), (order, mode)
        finally:
            oops.set_tuning(oops.TUNE_GW_TILE_ORDER, default_order); oops.set_tuning(oops.TUNE_GW_MARCHER, -1)
    # accumulating camera by camera == one call (torch.max over cameras, octree/extraction.py:206-212)
    acc = None
    for c in cams:
        acc = oops.grid_weight_render(torch.from_numpy(sigma).to(dev), reso, torch.from_numpy(c[None]).to(dev), fx, fx, W, H,
                                      oops.render_opts(1e-3), t.offset, t.invradius, grid_weight=acc)
    assert torch.equal(acc, got)


@pytest.mark.parametrize("K", [1, 4, 9, 16, 25])
def test_octree_render_matches_oracle(K):
    oops = _oops(); dev = _gpu()
    t = _random_tree(3, 10 + K, K)
    view, keep = _device_tree(t, dev)
    W, H, fx = 14, 10, 13.0
    for theta, phi, fast in ((20.0, 30.0, False), (250.0, -5.0, True)):
        c2w = _pose(theta, phi)
        opt = T.RenderOptions.for_renderer(1e-3, fast)
        want = T.render_persp(t, c2w, W, H, fx, opt)
        got = oops.octree_render_persp(view, torch.from_numpy(c2w).to(dev), W, H, fx,
                                       oops.render_opts(1e-3, 1.0, float(opt.sigma_thresh), float(opt.stop_thresh)))
        assert got.shape == (H, W, 3)
        assert float(np.abs(want - 1.0).max()) > 0.2                 # the view actually sees the tree
        close(f"SH{K} image fast={fast}", got, torch.from_numpy(want), rtol=0, atol=2e-5)
    # explicit rays (origins inside and outside the volume, one that misses)
    rs = np.random.RandomState(K)
    o = np.concatenate([rs.randn(20, 3) * 3.0, rs.rand(4, 3) * 0.5, [[9.0, 9.0, 9.0]]]).astype(f32)
    d = (-o + rs.randn(25, 3) * 0.4).astype(f32)
    d[-1] = [1.0, 0.0, 0.0]
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    opt = T.RenderOptions(2e-3, background_brightness=0.5)
    want = np.stack([T.render_ray(t, oo, dd, dd, opt) for oo, dd in zip(o, d)])
    to = lambda a: torch.from_numpy(a).to(dev)
    got = oops.octree_render_rays(view, to(o), to(d), to(d), oops.render_opts(2e-3, 0.5))
    close(f"SH{K} rays", got, torch.from_numpy(want), rtol=0, atol=2e-5)
    assert np.allclose(want[-1], 0.5) and torch.allclose(got[-1].cpu(), torch.full((3,), 0.5))


def test_work_counters_match_the_oracle_march():
    """pxo_octree_count_work / pxo_grid_weight_count_work (the roofline pass of scripts/octree_bench.py) repeat the kernels'
    march and count; the oracle's literal march (march_tree / grid_weight_render's loop) gives the same integers: rays that
    enter the volume, samples, samples above the sigma threshold (with the early stop of the `fast` preset), distinct leaves;
    child-pointer loads lie between one per sample (full path reuse) and depth + 1 per sample (none)."""
    oops = _oops(); dev = _gpu()
    t = _random_tree(3, 21, 4)
    view, keep = _device_tree(t, dev)
    W, H, fx = 14, 10, 13.0
    flat = t.data.reshape(-1, t.data_dim)
    for theta, phi, fast in ((20.0, 30.0, False), (250.0, -5.0, True)):
        c2w = _pose(theta, phi)
        opt = T.RenderOptions.for_renderer(1e-3, fast)
        rays = samples = shaded = 0
        leaves = set()
        for iy in range(H):
            for ix in range(W):
                o, d = T.cam2world_ray(ix, iy, c2w, W, H, fx, fx)
                seq = T.march_tree(t, o, d, opt)
                if seq is None:
                    continue
                rays += 1
                light = f32(1.0)
                for leaf, dtw in seq:
                    samples += 1
                    sg = flat[leaf][-1]
                    if sg > opt.sigma_thresh:
                        shaded += 1
                        leaves.add(leaf)
                        light = f32(light * f32(np.exp(f32(-dtw * sg), dtype=f32)))
                        if light <= opt.stop_thresh:
                            break
        got = oops.octree_count_work(view, torch.from_numpy(c2w).to(dev), W, H, fx,
                                     oops.render_opts(1e-3, 1.0, float(opt.sigma_thresh), float(opt.stop_thresh)))
        assert (got["rays"], got["samples"], got["shaded_samples"], got["distinct_leaves"]) == (rays, samples, shaded, len(leaves)), \
            (got, rays, samples, shaded, len(leaves))
        assert samples <= got["child_loads"] <= samples * 4 and samples > 500
    # the weight mask's march on a dense grid
    reso = 16
    sigma = ((np.random.RandomState(0).rand(reso, reso, reso) - 0.6) * 30.0).astype(f32)
    cams = np.stack([_pose(20.0, 30.0), _pose(200.0, -10.0)])
    opt = T.RenderOptions(step_size=1e-3)
    rays = samples = occ = 0
    seen = np.zeros_like(sigma, dtype=bool)
    for c in cams:
        for iy in range(11):
            for ix in range(13):
                origin, direction = T.cam2world_ray(ix, iy, c, 13, 11, 14.0, 14.0)
                o, d, invdir, delta_scale = T._to_tree_ray(origin, direction, t.offset, t.invradius)
                tmin, tmax = T._dda_unit(o, invdir)
                if tmax < 0 or tmin > tmax:
                    continue
                rays += 1
                tt, light = tmin, f32(1.0)
                while tt < tmax:
                    pos = np.clip(np.array([f32(o[a] + f32(tt * d[a])) for a in range(3)], f32), f32(0.0), f32(1.0 - 1e-6)).astype(f32)
                    pos = (pos * f32(reso)).astype(f32)
                    u = np.floor(pos).astype(np.int64)
                    s0, s1 = T._dda_unit((pos - u.astype(f32)).astype(f32), invdir)
                    delta_t = f32(f32(f32(s1 - s0) / f32(reso)) + opt.step_size)
                    samples += 1
                    sg = sigma[u[0], u[1], u[2]]
                    if sg > opt.sigma_thresh:
                        occ += 1
                        seen[u[0], u[1], u[2]] = True
                        light = f32(light * f32(np.exp(f32(-f32(delta_t * delta_scale) * sg), dtype=f32)))
                        if light <= opt.stop_thresh:
                            break
                    tt = f32(tt + delta_t)
    got = oops.grid_weight_count_work(torch.from_numpy(sigma).to(dev), reso, torch.from_numpy(cams).to(dev), 14.0, 14.0, 13, 11,
                                      oops.render_opts(1e-3), t.offset, t.invradius)
    assert (got["rays"], got["samples"], got["occupied_samples"], got["distinct_voxels"]) == (rays, samples, occ, int(seen.sum())), \
        (got, rays, samples, occ, int(seen.sum()))


@pytest.fixture(params=[0, 1], ids=["wave_per_sample", "ray_parallel"])
def bwd_update(request):
    """Both forms of the backward renderer's cache update (PXO_TUNE_BWD_UPDATE), each held to the oracle."""
    oops = _oops()
    default = oops.get_tuning(oops.TUNE_BWD_UPDATE)
    oops.set_tuning(oops.TUNE_BWD_UPDATE, request.param)
    yield request.param
    oops.set_tuning(oops.TUNE_BWD_UPDATE, default)


@pytest.mark.parametrize("K", [4, 16, 25])
def test_octree_render_gradient_matches_oracle(K, bwd_update):
    oops = _oops(); dev = _gpu()
    t = _random_tree(2, 30 + K, K, p=0.3)
    view, (child, data) = _device_tree(t, dev)
    rs = np.random.RandomState(K + 1)
    o = (rs.randn(24, 3) * 3.0).astype(f32)
    d = (-o + rs.randn(24, 3) * 0.3).astype(f32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    g = rs.randn(24, 3).astype(f32)
    opt = T.RenderOptions(1e-3)
    dd = torch.tensor(t.data.astype(np.float64), requires_grad=True)
    out = T.render_rays_torch(t, dd, o, d, d, opt)
    (out * torch.from_numpy(g.astype(np.float64))).sum().backward()
    to = lambda a: torch.from_numpy(a).to(dev)
    grad = torch.zeros_like(data)
    oops.octree_render_rays_bwd(view, to(o), to(d), to(d), oops.render_opts(1e-3), to(g), grad)
    want = dd.grad.float()
    assert float(want.abs().max()) > 1e-3
    close(f"SH{K} d/d data", grad, want, rtol=2e-3, atol=2e-6 * float(want.abs().max()) + 1e-7)
    # accumulation: a second call doubles the gradient; handing over the forward result skips one march
    fwd = oops.octree_render_rays(view, to(o), to(d), to(d), oops.render_opts(1e-3))
    oops.octree_render_rays_bwd(view, to(o), to(d), to(d), oops.render_opts(1e-3), to(g), grad, out_rgb=fwd)
    close("accumulated", grad, 2 * want, rtol=2e-3, atol=4e-6 * float(want.abs().max()) + 1e-7)
    # camera mode + the autograd bridge used by a reference-style training loop
    from plenoctree_amd.octree import svox
    W, H, fx = 12, 9, 11.0
    c2w = _pose(40.0, 25.0)
    rays = [T.cam2world_ray(ix, iy, c2w, W, H, fx, fx) for iy in range(H) for ix in range(W)]
    ro = np.stack([r[0] for r in rays]); rd = np.stack([r[1] for r in rays])
    gt = rs.rand(H, W, 3).astype(f32)
    dd = torch.tensor(t.data.astype(np.float64), requires_grad=True)
    im = T.render_rays_torch(t, dd, ro, rd, rd, opt).reshape(H, W, 3)
    mse = ((im.clamp(0.0, 1.0) - torch.from_numpy(gt.astype(np.float64))) ** 2).mean()
    mse.backward()
    radius = 0.5 / t.invradius
    host = svox.N3Tree(N=2, data_dim=t.data_dim, depth_limit=2, radius=radius, center=(1.0 - 2.0 * t.offset) * radius,
                       data_format=f"SH{K}", map_location=dev)
    host.child, host.parent_depth = child, torch.from_numpy(t.parent_depth).to(dev)
    host.data = data.clone().requires_grad_(True)
    host.level_nodes = np.bincount(t.parent_depth[:, 1]).tolist()
    r = svox.VolumeRenderer(host, step_size=1e-3)
    im_d = r.render_persp(torch.from_numpy(c2w), width=W, height=H, fx=fx)
    close("bridge image", im_d, im.detach().float(), rtol=0, atol=2e-5)
    loss = ((im_d.clamp(0.0, 1.0) - to(gt)) ** 2).mean()
    loss.backward()
    assert abs(float(loss) - float(mse)) < 1e-6
    close("bridge grad", host.data.grad, dd.grad.float(), rtol=2e-3, atol=2e-6 * float(dd.grad.abs().max()) + 1e-9)
    # the fused loss-gradient kernel agrees with torch's clamp + mse backward
    sse, gi = oops.image_mse(im_d.detach(), to(gt))
    assert abs(float(sse) / im_d.numel() - float(loss)) < 1e-6
    ref_in = im_d.detach().clone().requires_grad_(True)
    ((ref_in.clamp(0.0, 1.0) - to(gt)) ** 2).mean().backward()
    close("image_mse grad", gi, ref_in.grad, rtol=1e-6, atol=1e-9)


@pytest.mark.parametrize("lanes", [4, 8, 16])
def test_octree_render_every_lanes_per_ray_template(lanes):
    """The renderer is instantiated for 4, 8 and 16 lanes per ray (the default launches use 4: backward = 4-lane march + 16-lane cooperative scatter);
    every instantiation is held to the same oracle bounds, forward and gradient, for every SH format."""
    oops = _oops(); dev = _gpu()
    try:
        oops.set_lanes_per_ray(lanes, lanes)
        for K in (1, 4, 9, 16, 25):
            t = _random_tree(3, 50 + K, K)
            view, (child, data) = _device_tree(t, dev)
            W, H, fx = 13, 9, 12.0
            c2w = _pose(35.0, 20.0)
            opt = T.RenderOptions(1e-3)
            want = T.render_persp(t, c2w, W, H, fx, opt)
            got = oops.octree_render_persp(view, torch.from_numpy(c2w).to(dev), W, H, fx, oops.render_opts(1e-3))
            close(f"SH{K} image, {lanes} lanes", got, torch.from_numpy(want), rtol=0, atol=2e-5)
            rs = np.random.RandomState(K + lanes)
            o = (rs.randn(16, 3) * 3.0).astype(f32)
            d = (-o + rs.randn(16, 3) * 0.3).astype(f32)
            d /= np.linalg.norm(d, axis=1, keepdims=True)
            g = rs.randn(16, 3).astype(f32)
            dd = torch.tensor(t.data.astype(np.float64), requires_grad=True)
            out = T.render_rays_torch(t, dd, o, d, d, opt)
            (out * torch.from_numpy(g.astype(np.float64))).sum().backward()
            to = lambda a: torch.from_numpy(a).to(dev)
            grad = torch.zeros_like(data)
            oops.octree_render_rays_bwd(view, to(o), to(d), to(d), oops.render_opts(1e-3), to(g), grad)
            wantg = dd.grad.float()
            close(f"SH{K} d/d data, {lanes} lanes", grad, wantg, rtol=2e-3, atol=2e-6 * float(wantg.abs().max()) + 1e-7)
    finally:
        oops.set_lanes_per_ray(0, 0)
    from plenoctree_amd import _lib
    with pytest.raises(_lib.PxoError, match="lanes"):
        oops.set_lanes_per_ray(5, 0)


def test_svox_call_surface_on_device():
    """The svox indexing surface the reference's extraction uses (octree/extraction.py:337-394,:503), call for call,
    on the device: `tree[grid].refine()` x depth, `tree.depths`, `tree[inds].sample(S)`, `tree[inds] = rgba`,
    `tree[:, -1:].relu_()`.  (tests/test_reference_drivers_cpu.py runs the reference's own step1/step2 on the same
    classes with oracle-backed kernels; /root/reference does not exist on this box.)"""
    oops = _oops(); dev = _gpu()
    from plenoctree_amd.octree import svox
    depth, K, center, radius = 3, 4, [0.1, 0.0, -0.2], [1.4, 1.5, 1.3]
    reso = 2 ** (depth + 1)
    mask = _mask(depth, 77, 0.1)
    want = T.build_from_mask(mask, depth, 3 * K + 1, center, radius)

    def new_tree():
        return svox.N3Tree(N=2, data_dim=3 * K + 1, init_refine=0, init_reserve=500000, geom_resize_fact=1.0,
                           depth_limit=depth, radius=radius, center=center, data_format=f"SH{K}", extra_data=None,
                           map_location=dev)

    tree = new_tree()
    assert tree.offset.device.type == "cpu" and torch.equal(tree.offset.cpu(), torch.from_numpy(want.offset))
    # leaf lookup against the oracle's descent, including points outside the volume (clamped) and on cell faces
    rs = np.random.RandomState(5)
    probe = np.concatenate([rs.uniform(-2.0, 2.0, (500, 3)), T.grid_points(reso, want.offset, want.invradius)[::7]]).astype(f32)
    fine = svox.N3Tree(N=2, data_dim=3 * K + 1, depth_limit=depth, radius=radius, center=center, data_format=f"SH{K}",
                       map_location=dev).refine_from_mask(torch.from_numpy(mask.astype(np.uint8)).to(dev))
    got = oops.tree_query(fine.child, torch.from_numpy(probe).to(dev), fine.offset, fine.invradius).cpu().numpy()
    ref = []
    for pnt in probe:
        n, i, j, k, _, _ = want.query(want.world2tree(pnt))
        ref.append(((n * 2 + i) * 2 + j) * 2 + k)
    assert np.array_equal(got, np.asarray(ref))
    # octree/extraction.py:337-350
    arr = (torch.arange(0, reso, dtype=torch.float32) + 0.5) / reso
    xx, yy, zz = [(arr - tree.offset.cpu()[a]) / tree.invradius.cpu()[a] for a in range(3)]
    grid = torch.stack(torch.meshgrid(xx, yy, zz, indexing="ij")).reshape(3, -1).T
    grid = grid[torch.from_numpy(mask.reshape(-1))].to(dev)
    for _ in range(depth - 1):
        tree[grid].refine()
    half = grid.shape[0] // 2                        # "do last layer separately" in chunks (:345-350)
    tree[grid[:half]].refine()
    tree[grid[half:]].refine()
    assert tree.max_depth == depth and tree.n_internal == want.n_internal
    # chunked refinement appends the last level chunk by chunk: same tree, node order of that level differs; one-shot
    # refinement reproduces the oracle's (and refine_from_mask's) arrays exactly
    one = new_tree()
    for _ in range(depth):
        one[grid].refine()
    assert np.array_equal(one.child.cpu().numpy(), want.child) and np.array_equal(one.parent_depth.cpu().numpy(), want.parent_depth)
    assert torch.equal(one.child, fine.child) and torch.equal(one.parent_depth, fine.parent_depth)
    assert np.array_equal(one.depths.cpu().numpy(), want.depths())
    for t in (tree, one):
        # :358-394
        leaf_mask = t.depths.cpu() == t.max_depth
        leaf_ind = torch.where(leaf_mask)[0]
        assert leaf_ind.numel() == 8 * int((want.parent_depth[:, 1] == depth).sum())
        S = 5
        for i in range(0, leaf_ind.size(0), 300):
            chunk_inds = leaf_ind[i:i + 300]
            points = t[chunk_inds].sample(S)
            assert points.shape == (chunk_inds.numel(), S, 3)
            packed = t._leaf_packed()[chunk_inds.to(dev)]
            again = oops.tree_query(t.child, points.view(-1, 3), t.offset, t.invradius).view(-1, S)
            assert torch.equal(again, packed[:, None].expand(-1, S))          # every sample lies inside its own leaf
            assert float(points.view(-1, S, 3).std(dim=1).min()) > 0          # and they are distinct
            rgba = torch.cat([points.mean(dim=1), points.new_full((points.shape[0], 3 * K - 3), 0.25),
                              points[:, 0, :1] - 0.1], -1)
            t[chunk_inds] = rgba
            assert torch.equal(t.data.data.view(-1, 3 * K + 1)[packed], rgba)
        flat = t.data.data.view(-1, 3 * K + 1)
        deep = t._leaf_packed()[leaf_ind.to(dev)]
        untouched = torch.ones(flat.shape[0], dtype=torch.bool, device=dev); untouched[deep] = False
        assert float(flat[untouched].abs().max()) == 0.0
        assert bool((flat[:, -1] < 0).any())
        t[:, -1:].relu_()
        assert float(t.data.data[..., -1].min()) == 0.0 and bool((t.data.data[..., 0] < 0).any())
    # both trees render the same image (same geometry, same leaf values up to the random samples -> compare structure only)
    assert tree.parameters()[0] is tree.data and tree.data.requires_grad


def test_sgd_step_matches_torch():
    oops = _oops(); dev = _gpu()
    gen = torch.Generator(device=dev).manual_seed(0)
    for mu, nesterov in ((0.0, False), (0.9, False), (0.9, True)):
        p = torch.randn(1000, device=dev, generator=gen)
        ref = torch.nn.Parameter(p.clone())
        opt = torch.optim.SGD([ref], lr=0.1, momentum=mu, nesterov=nesterov)
        buf = torch.zeros_like(p) if mu else None
        for step in range(3):
            g = torch.randn(1000, device=dev, generator=gen)
            ref.grad = g.clone()
            opt.step()
            oops.sgd_step(p, g, 0.1, mu, nesterov, buf, first_step=step == 0)
            close(f"sgd mu={mu} nesterov={nesterov} step {step}", p, ref.data, rtol=1e-6, atol=1e-6)   # fma vs mul+add


def test_octree_error_paths():
    oops = _oops(); dev = _gpu()
    from plenoctree_amd._lib import PxoError
    t = _random_tree(1, 1, 4)
    view, keep = _device_tree(t, dev)
    c2w = torch.from_numpy(_pose(10.0, 10.0)).to(dev)
    with pytest.raises(PxoError, match="step_size"):
        oops.octree_render_persp(view, c2w, 4, 4, 5.0, oops.render_opts(0.0))
    bad = oops.tree_view(keep[0], keep[1], t.offset, t.invradius); bad.basis_dim = 5
    with pytest.raises(PxoError, match="basis_dim"):
        oops.octree_render_persp(bad, c2w, 4, 4, 5.0, oops.render_opts(1e-3))
    with pytest.raises(PxoError, match="uint8"):
        oops.tree_from_mask(torch.zeros(8 ** 3, device=dev), 2)
    with pytest.raises(PxoError, match="data_dim"):
        oops.tree_view(keep[0], torch.zeros(3, 2, 2, 2, 12, device=dev), t.offset, t.invradius)
    with pytest.raises(PxoError, match="depth"):
        oops.tree_workspace_bytes(11)
    # the gradient pass marches exactly: an early-stopped forward image is rejected at the C ABI
    fast = oops.render_opts(1e-3, 1.0, 1e-2, 1e-2)
    img = oops.octree_render_persp(view, c2w, 4, 4, 5.0, fast)
    with pytest.raises(PxoError, match="stop_thresh"):
        oops.octree_render_persp_bwd(view, c2w, 4, 4, 5.0, fast, torch.ones_like(img), torch.zeros_like(keep[1]), out_rgb=img)
    # the weight-mask work counters repeat the power-of-two kernels' march only
    with pytest.raises(PxoError, match="power of two"):
        oops.grid_weight_count_work(torch.zeros(12 ** 3, device=dev), 12, c2w[None], 5.0, 5.0, 4, 4,
                                    oops.render_opts(1e-3), t.offset, t.invradius)
    with pytest.raises(PxoError, match="pxo_octree_set_tuning"):
        oops.set_tuning(oops.TUNE_BWD_UPDATE, 2)


def _write_checkpoint(tmp_path, args, flat):
    from plenoctree_amd.nerf_sh.nerf import checkpoints, models
    model, _ = models.construct_nerf(args, torch.device("cuda:0"))
    state = models.TrainState(model.cfg, flat.to("cuda:0"))
    checkpoints.save_checkpoint(str(tmp_path), state, step=0)
    return model, state


def test_extraction_pipeline_end_to_end(tmp_path):
    """octree.extraction -> octree.evaluation -> octree.optimization through their CLIs on a model whose density
    field is known to be non-trivial; the extracted tree's leaf data is the mean of the network output at the
    leaf's own sample points, and its render approximates the NeRF render of the same model."""
    oops = _oops(); dev = _gpu()
    from plenoctree_amd import ops
    from plenoctree_amd.nerf_sh.nerf import models, utils
    from plenoctree_amd.octree import evaluation, extraction, optimization, svox
    cfg_path = os.path.join(str(tmp_path), "tiny.yaml")
    with open(cfg_path, "w") as f:
        f.write("dataset: synthetic\nfactor: 16\nnum_coarse_samples: 64\nnum_fine_samples: 128\nuse_viewdirs: false\n"
                "white_bkgd: true\nbatch_size: 1024\nsh_deg: 3\nrandomized: true\n")
    args = utils.define_flags().parse_args(["--train_dir", str(tmp_path), "--config", cfg_path])
    utils.update_flags(args)
    flat = make_params(O.Cfg(), seed=11, bias_scale=0.2)
    model, state = _write_checkpoint(tmp_path, args, flat)
    out = os.path.join(str(tmp_path), "tree.npz")
    common = ["--train_dir", str(tmp_path), "--config", cfg_path]
    for mode in ("sigma", "weight"):
        tree = extraction.main(common + ["--output", out, "--init_grid_depth", "4", "--masking_mode", mode,
                                         "--samples_per_cell", "8", "--renderer_step_size", "1e-3", "--eval", "false"])
        assert tree.max_depth == 4 and tree.n_internal > 20
        assert bool((tree.data[..., -1] >= 0).all())
    # leaf data == mean over the leaf's samples of the network output (same Philox seed)
    node0, count = tree.max_depth_nodes()
    pts = tree.sample_max_depth_cells(8, first=0, count=count, seed=args.seed)
    rgb, sigma = model.eval_points_raw(state, pts.view(-1, 3))
    want = torch.cat([rgb, sigma], -1).reshape(-1, 8, 49).mean(1)
    want[:, -1].clamp_(min=0)
    # chunk boundaries reuse stream ids per `first`, so compare the first chunk only
    close("leaf data", tree.max_depth_data()[: 64 * 8], want[: 64 * 8], rtol=1e-5, atol=1e-6)
    # the same extraction with the opt-in split-precision MLP forward (--mlp_precision bf16x3): same tree, same leaf data
    # to the split-precision error (~1e-5 on values of order 1)
    tree_x3 = extraction.main(common + ["--output", os.path.join(str(tmp_path), "tree_x3.npz"), "--init_grid_depth", "4",
                                        "--masking_mode", "weight", "--samples_per_cell", "8", "--renderer_step_size", "1e-3",
                                        "--eval", "false", "--mlp_precision", "bf16x3"])
    assert torch.equal(tree_x3.child, tree.child)
    close("leaf data, bf16x3 extraction", tree_x3.data.data, tree.data.data, rtol=1e-4, atol=2e-4)
    # the saved file loads and renders; early-stop and exact renders agree to the thresholds
    loaded = svox.N3Tree.load(out, map_location=dev)
    assert loaded.n_internal == tree.n_internal and torch.equal(loaded.child, tree.child)
    vid = os.path.join(str(tmp_path), "eval.gif")
    psnr = evaluation.main(common + ["--input", out, "--renderer_step_size", "1e-3", "--approx_eval_skip", "1",
                                     "--write_vid", vid])
    assert np.isfinite(psnr) and os.path.exists(vid)
    with pytest.raises(ValueError):        # only GIF output is built (no imageio / ffmpeg): any other extension is refused
        evaluation.main(common + ["--input", out, "--renderer_step_size", "1e-3", "--approx_eval_skip", "1",
                                  "--write_vid", os.path.join(str(tmp_path), "eval.mp4")])
    # --z_min / --z_max (extraction.py:91-100, :298-301): grid planes outside the world-z range never enter the tree, the
    # rest of the mask is untouched; the reference's command line spelling (--is_jaxnerf_ckpt etc.) parses
    tz = extraction.main(common + ["--output", os.path.join(str(tmp_path), "tree_z.npz"), "--init_grid_depth", "4",
                                   "--masking_mode", "sigma", "--samples_per_cell", "8", "--eval", "false", "--z_max", "0.2",
                                   "--z_min", "-0.9", "--is_jaxnerf_ckpt", "--max_refine_prop", "0.5",
                                   "--projection_samples", "10000"])
    reso = 32
    grid = (torch.stack(torch.meshgrid(*[(torch.arange(reso, device=dev) + 0.5) / reso] * 3, indexing="ij"), -1).reshape(-1, 3)
            - 0.5) * 3.0
    leaves = oops.tree_query(tz.child, grid, tz.offset, tz.invradius)
    depth_of = tz.parent_depth[leaves // 8, 1].view(reso, reso, reso)
    zw = ((torch.arange(reso, device=dev) + 0.5) / reso - 0.5) * 3.0
    outside = (zw > 0.2) | (zw < -0.9)
    assert int((depth_of[:, :, outside] == 4).sum()) == 0 and int((depth_of[:, :, ~outside] == 4).sum()) > 0
    # a few optimisation epochs on 50x50 images improve the training PSNR
    hist = optimization.main(common + ["--input", out, "--output", os.path.join(str(tmp_path), "tree_opt.npz"),
                                       "--num_epochs", "3", "--val_interval", "1", "--renderer_step_size", "1e-3",
                                       "--lr", "2e4", "--continue_on_decrease", "--render_interval", "100"])
    train_psnrs = [h[1] for h in hist if h[1] is not None]
    assert len(train_psnrs) == 3 and train_psnrs[-1] > train_psnrs[0], hist
    assert os.path.exists(os.path.join(os.path.splitext(out)[0] + "_render", "0000_0000.png"))    # optimization.py:163,205


def test_octree_full_size_properties():
    """Reference sizes (depth 8 tree over a 512^3 mask, 800x800 view): properties that need no oracle --
    camera mode == explicit-ray mode, bit-identical repeats, early-stop within its thresholds of the exact
    render, gradient linear in grad_out and zero where nothing was seen."""
    oops = _oops(); dev = _gpu()
    from plenoctree_amd import ops
    from plenoctree_amd.octree import svox
    depth, reso, K = 8, 512, 16
    ax = ((torch.arange(reso, device=dev, dtype=torch.float32) + 0.5) / reso - 0.5) * 3.0
    d2 = ax[:, None, None] ** 2 + ax[None, :, None] ** 2 + (ax[None, None, :] * 1.3) ** 2
    mask = ((d2.sqrt() - 0.8).abs() < 0.02).to(torch.uint8).reshape(-1)         # thin ellipsoidal shell
    tree = svox.N3Tree(N=2, data_dim=3 * K + 1, depth_limit=depth, radius=1.5, center=[0, 0, 0], data_format="SH16",
                       map_location=dev)
    tree.refine_from_mask(mask)
    assert tree.max_depth == depth and tree.n_internal > 100000
    leaf = tree.max_depth_data()
    g = torch.Generator(device=dev).manual_seed(0)
    leaf.copy_(torch.randn(leaf.shape, device=dev, generator=g) * 0.5)
    leaf[:, -1] = (torch.rand(leaf.shape[0], device=dev, generator=g) - 0.3) * 60.0
    W = H = 800
    focal = 0.5 * W / math.tan(0.5 * 0.6911112)
    c2w = torch.from_numpy(_pose(35.0, 25.0, 4.0311)).to(dev)
    r = svox.VolumeRenderer(tree, step_size=1e-4)
    im = r.render_persp(c2w, width=W, height=H, fx=focal)
    assert im.shape == (H, W, 3) and bool(torch.isfinite(im).all())
    seen = (im - 1.0).abs().amax(dim=-1) > 1e-3
    assert 0.05 < float(seen.float().mean()) < 0.6                            # the shell covers part of the view
    assert torch.equal(im, r.render_persp(c2w, width=W, height=H, fx=focal))  # deterministic
    o, d, v = ops.generate_rays(c2w, W, H, focal)
    im_rays = r.forward(o, v, v).reshape(H, W, 3)                             # unit directions
    # the two ray set-ups round differently (rotate-then-normalise vs normalise-then-rotate), so a few rays take a
    # sample on the other side of a cell face of this thin, dense shell: identical up to rare, bounded outliers
    diff = (im_rays - im).abs()
    assert float((diff > 2e-5).float().mean()) < 5e-3 and float(diff.max()) < 0.05 and float(diff.mean()) < 2e-6, (
        float((diff > 2e-5).float().mean()), float(diff.max()), float(diff.mean()))
    with torch.no_grad():
        fast = r.render_persp(c2w, width=W, height=H, fx=focal, fast=True)
    assert float((fast - im).abs().max()) < 0.03                              # sigma/stop thresholds of 1e-2
    if tree.data.requires_grad:       # the early-stopping preset is not differentiable: loud, not a detached image
        with pytest.raises(oops.PxoError):
            r.render_persp(c2w, width=W, height=H, fx=focal, fast=True)
    # gradient: linear in grad_out, confined to leaves that rays reached
    g1, g2 = torch.randn(H, W, 3, device=dev, generator=g), torch.randn(H, W, 3, device=dev, generator=g)
    opts = r._opts(False)
    grads = []
    for go in (g1, g2, g1 + g2):
        gd = torch.zeros_like(tree.data)
        oops.octree_render_persp_bwd(tree.view(), c2w, W, H, focal, opts, go.contiguous(), gd)
        grads.append(gd)
    scale = float(grads[2].abs().max())
    assert scale > 0
    assert float((grads[0] + grads[1] - grads[2]).abs().max()) < 2e-4 * scale
    node0, _ = tree.max_depth_nodes()
    assert not bool(grads[2][:node0].any())                                   # shallower leaves hold sigma = 0
    hit = grads[2][node0:].abs().amax(dim=-1) > 0
    assert 0.01 < float(hit.float().mean()) < 0.9
    # a pixel that misses the volume gets no gradient and the background colour
    miss = ~seen
    assert bool(miss.any()) and float((im[miss] - 1.0).abs().max()) <= 1e-3


@pytest.mark.parametrize("K", [16, 25])
def test_render_kernel_variants_agree_at_image_size(K):
    """The lane-count variants are different kernels (forward: channel-aligned 16-byte reads at 4 lanes, index-ordered at
    8 / 16; backward: 4-lane march + 16-lane cooperative scatter against the plain 8- / 16-lane kernels).  The oracle
    comparisons above hold each of them on a handful of rays; here they are held against each other on a 401 x 397 image
    (ragged patches on both borders) of a depth-6 tree: images to the oracle tolerance, gradients (float atomics: the order
    of the adds is not fixed) to 1e-5 relative, repeat launches of the forward bit-identical, and the gradient launch
    that re-marches for the colour equal to the one that is handed the forward image."""
    oops = _oops(); dev = _gpu()
    t = _random_tree(6, 900 + K, K, p=0.05)
    view, (child, data) = _device_tree(t, dev)
    W, H, fx = 401, 397, 420.0
    c2w = torch.from_numpy(_pose(40.0, 25.0)).to(dev)
    opt = oops.render_opts(1e-3)
    gout = torch.randn(H, W, 3, device=dev, generator=torch.Generator(device=dev).manual_seed(K))
    imgs, grads = {}, {}
    try:
        for lanes in (4, 8, 16):
            oops.set_lanes_per_ray(lanes, lanes)
            imgs[lanes] = oops.octree_render_persp(view, c2w, W, H, fx, opt)
            assert torch.equal(imgs[lanes], oops.octree_render_persp(view, c2w, W, H, fx, opt))
            g = torch.zeros_like(data)
            oops.octree_render_persp_bwd(view, c2w, W, H, fx, opt, gout, g, out_rgb=imgs[lanes])
            grads[lanes] = g
        g2 = torch.zeros_like(data)
        oops.set_lanes_per_ray(4, 4)
        oops.octree_render_persp_bwd(view, c2w, W, H, fx, opt, gout, g2)          # two marches, no forward image
        # the 4-lane backward's cache: both update forms, every cache size
        default_upd, default_rows = oops.get_tuning(oops.TUNE_BWD_UPDATE), oops.get_tuning(oops.TUNE_BWD_CACHE_ROWS)
        try:
            for upd in (0, 1):
                for rows in (4, 16, 64):
                    oops.set_tuning(oops.TUNE_BWD_UPDATE, upd); oops.set_tuning(oops.TUNE_BWD_CACHE_ROWS, rows)
                    g = torch.zeros_like(data)
                    oops.octree_render_persp_bwd(view, c2w, W, H, fx, opt, gout, g, out_rgb=imgs[4])
                    grads[("cache", upd, rows)] = g
        finally:
            oops.set_tuning(oops.TUNE_BWD_UPDATE, default_upd); oops.set_tuning(oops.TUNE_BWD_CACHE_ROWS, default_rows)
    finally:
        oops.set_lanes_per_ray(0, 0)
    for key, g in grads.items():
        if isinstance(key, tuple):
            rel = float((g.double() - grads[16].double()).norm() / grads[16].double().norm())
            assert rel < 1e-5, (key, rel)
    assert float(imgs[16].std()) > 0.05                                              # a real image, not background
    for lanes in (4, 8):
        close(f"SH{K} image {lanes} vs 16 lanes", imgs[lanes], imgs[16], rtol=0, atol=2e-5)
        rel = float((grads[lanes].double() - grads[16].double()).norm() / grads[16].double().norm())
        assert rel < 1e-5, (lanes, rel)
    assert float(grads[16].abs().max()) > 0
    rel = float((g2.double() - grads[4].double()).norm() / grads[4].double().norm())
    assert rel < 1e-5, rel
