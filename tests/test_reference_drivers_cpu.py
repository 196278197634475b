"""The reference's OWN octree drivers, unmodified, running on plenoctree_amd.octree.svox.

`/root/reference/octree/{extraction,optimization,evaluation,compression}.py` are imported as they are, with
`sys.modules["svox"]` / `["svox.helpers"]` pointing at plenoctree_amd.octree.svox (svox.install_as_svox()), and stub
modules for what the container lacks and that carry no arithmetic of the path (absl flag registration, cv2, imageio,
lpips).  This pins the svox CALL SURFACE -- names, argument order, indexing forms, attribute types, npz keys -- against
the code that consumes it:

    extraction.step1 / step2      tree.offset.cpu(), tree[grid].refine(), tree.max_depth, tree.depths,
                                  tree[inds].sample(S), tree[inds] = rgba, tree.data_format.format, _C.RenderOptions /
                                  CameraSpec / grid_weight_render                      (octree/extraction.py:181-394)
    optimization.main             svox.N3Tree.load, svox.VolumeRenderer(...).render_persp(..., cuda=True), t.parameters()
                                  through torch.optim.SGD, mse.backward() into t.data.grad, t.clone(device='cpu'),
                                  best_t.save                                           (octree/optimization.py:136-243)
    evaluation.main / eval_octree render_persp(..., fast=...) + im.clamp_()             (octree/nerf/utils.py:448-497)
    compression.main --noquant    the npz key list                                      (octree/compression.py:76-86,138)

It runs on the CPU: the HIP entry points behind svox (leaf lookup, leaf sampling, renderer forward/backward, weight
mask) are replaced by stand-ins backed by oracle/octree_oracle.py, exactly as tests/test_distributed_octree_cpu.py
does; tests/test_gpu_octree.py::test_svox_call_surface_on_device makes the same calls against the real kernels
(the reference tree itself does not exist on the GPU box).  What this does NOT pin is svox's arithmetic (the oracle is a restatement, see its header).

Needs /root/reference (present in the authoring container only) -- skipped elsewhere.
"""
import importlib
import os
import sys
import types

import numpy as np
import pytest
import torch

from oracle import octree_oracle as T

REF = "/root/reference"
# The reference's driver modules are imported and executed inside this process (that is the point of the test: its own code
# consumes our svox mirror).  PXO_SKIP_REFERENCE_DRIVERS=1 opts out where running reference code is not wanted.
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "octree")) or os.environ.get("PXO_SKIP_REFERENCE_DRIVERS") == "1",
                                reason="needs /root/reference (and PXO_SKIP_REFERENCE_DRIVERS unset)")

K, D = 4, 13                    # SH4: 3*4 + 1 channels
W, H, FX = 7, 5, 6.0
f32 = np.float32


# ------------------------------------------------------------------------------------------------------------------
# environment: stubs + svox -> ours + oracle-backed stand-ins for the HIP entry points
# ------------------------------------------------------------------------------------------------------------------
def _install_stubs():
    absl, flags, app = types.ModuleType("absl"), types.ModuleType("absl.flags"), types.ModuleType("absl.app")
    flags.FLAGS = types.SimpleNamespace()

    def define(name, default=None, *a, **k):
        setattr(flags.FLAGS, name, default)

    for n in ("DEFINE_string", "DEFINE_integer", "DEFINE_float", "DEFINE_bool", "DEFINE_enum", "DEFINE_boolean"):
        setattr(flags, n, define)
    app.run = lambda main: None
    absl.flags, absl.app = flags, app
    imageio = types.ModuleType("imageio")
    imageio.imwrite = imageio.mimwrite = lambda *a, **k: None
    lpips = types.ModuleType("lpips")

    class LPIPS:                                  # pretrained VGG weights cannot be fetched: constant 0
        def __init__(self, net="vgg"):
            pass

        def eval(self):
            return self

        def to(self, device):
            return self

        def __call__(self, a, b, normalize=True):
            return torch.zeros(1)

    lpips.LPIPS = LPIPS
    sys.modules.update({"absl": absl, "absl.flags": flags, "absl.app": app, "cv2": types.ModuleType("cv2"),
                        "imageio": imageio, "lpips": lpips})
    return flags.FLAGS


def _oracle_tree_of(view):
    """oracle Tree over the arrays of an oops.tree_view stand-in."""
    t = object.__new__(T.Tree)
    t.child = view.child.cpu().numpy()
    t.data = view.data.detach().cpu().numpy().astype(f32)
    t.data_dim = t.data.shape[-1]
    t.offset = np.asarray([float(v) for v in view.offset], f32)
    t.invradius = np.asarray([float(v) for v in view.invradius], f32)
    t.depth_limit = 10
    return t


def _rays(c2w):
    rays = [T.cam2world_ray(ix, iy, np.asarray(c2w, f32), W, H, FX, FX) for iy in range(H) for ix in range(W)]
    return np.stack([r[0] for r in rays]), np.stack([r[1] for r in rays])


def _patch_ops(monkeypatch, log):
    from plenoctree_amd import octree_ops as oops

    def render_opts(step_size=1e-3, background_brightness=1.0, sigma_thresh=0.0, stop_thresh=0.0):
        return types.SimpleNamespace(step_size=step_size, background_brightness=background_brightness,
                                     sigma_thresh=sigma_thresh, stop_thresh=stop_thresh)

    def _opt(o):
        return T.RenderOptions(o.step_size, o.background_brightness, o.sigma_thresh, o.stop_thresh)

    def tree_view(child, data, offset, invradius):
        return types.SimpleNamespace(child=child, data=data, offset=offset, invradius=invradius)

    def octree_render_persp(view, c2w, width, height, fx, opts, fy=None):
        assert (width, height) == (W, H)
        log.append(("render", float(opts.stop_thresh)))
        t = _oracle_tree_of(view)
        ro, rd = _rays(c2w.cpu().numpy())
        return T.render_rays_torch(t, view.data.detach().double(), ro, rd, rd, _opt(opts)).reshape(H, W, 3).float()

    def octree_render_persp_bwd(view, c2w, width, height, fx, opts, grad_out, grad_data, fy=None, out_rgb=None):
        log.append(("render_bwd", out_rgb is not None))
        t = _oracle_tree_of(view)
        with torch.enable_grad():                 # called from inside autograd.Function.backward (grad mode off there)
            d = view.data.detach().double().requires_grad_(True)
            ro, rd = _rays(c2w.cpu().numpy())
            im = T.render_rays_torch(t, d, ro, rd, rd, _opt(opts)).reshape(H, W, 3)
            (im * grad_out.double()).sum().backward()
        grad_data += d.grad.float()
        return grad_data

    def tree_query(child, points, offset, invradius):
        t = _oracle_tree_of(types.SimpleNamespace(child=child, data=torch.zeros(1, 2, 2, 2, D), offset=offset,
                                                  invradius=invradius))
        out = []
        for p in points.reshape(-1, 3).cpu().numpy().astype(f32):
            n, i, j, k, _, _ = t.query(t.world2tree(p))
            out.append(((n * 2 + i) * 2 + j) * 2 + k)
        return torch.tensor(out, dtype=torch.int64)

    def tree_sample_leaves(parent_depth, packed, samples, offset, invradius, u=None, seed=0, stream_id=0):
        t = object.__new__(T.Tree)
        t.parent_depth = parent_depth.cpu().numpy()
        t.child = np.zeros((t.parent_depth.shape[0], 2, 2, 2), np.int32)
        pk = packed.cpu().numpy()
        leaves = np.stack([pk >> 3, (pk >> 2) & 1, (pk >> 1) & 1, pk & 1], 1)
        corner, side = T.leaf_corners(t, leaves)
        g = torch.Generator().manual_seed(1234 + stream_id)
        uu = torch.rand(len(pk), samples, 3, generator=g, dtype=torch.float64).numpy()
        p_tree = corner[:, None, :] + uu * side[:, None, None]
        off = np.asarray([float(v) for v in offset]); inv = np.asarray([float(v) for v in invradius])
        pts = torch.from_numpy(((p_tree - off) / inv).astype(f32))
        log.append(("sample", packed.clone(), pts.clone()))
        return pts

    def tree_relu_sigma(data):
        data[..., -1].clamp_(min=0.0)

    def grid_weight_render(sigma_grid, reso, c2w_all, fx, fy, width, height, opts, offset, invradius, grid_weight=None):
        log.append(("grid_weight", c2w_all.shape[0]))
        w = None
        off = np.asarray([float(v) for v in offset], f32); inv = np.asarray([float(v) for v in invradius], f32)
        for c in c2w_all.cpu().numpy():
            w = T.grid_weight_render(sigma_grid.reshape(reso, reso, reso).cpu().numpy().astype(f32), c, width, height, fx,
                                     _opt(opts), off, inv, weight=w)
        return torch.from_numpy(np.asarray(w, f32).reshape(-1))

    for name, fn in dict(render_opts=render_opts, tree_view=tree_view, octree_render_persp=octree_render_persp,
                         octree_render_persp_bwd=octree_render_persp_bwd, tree_query=tree_query,
                         tree_sample_leaves=tree_sample_leaves, tree_relu_sigma=tree_relu_sigma,
                         grid_weight_render=grid_weight_render).items():
        monkeypatch.setattr(oops, name, fn)


@pytest.fixture()
def ref(monkeypatch, tmp_path):
    """(FLAGS, log): the reference tree importable, svox = ours, HIP entry points = oracle stand-ins."""
    def purge():            # the reference's modules are imported afresh for every test (they bind FLAGS at import)
        for name in [n for n in sys.modules if n.split(".")[0] == "octree"]:
            del sys.modules[name]

    saved = dict(sys.modules)
    purge()
    FLAGS = _install_stubs()
    from plenoctree_amd.octree import svox
    svox.install_as_svox()
    monkeypatch.syspath_prepend(REF)
    monkeypatch.setattr(torch.Tensor, "cuda", lambda self, *a, **k: self, raising=False)
    log = []
    _patch_ops(monkeypatch, log)
    cfg = tmp_path / "stub_config"
    (tmp_path / "stub_config.yaml").write_text("white_bkgd: true\n")
    FLAGS._config_path = str(cfg)
    yield FLAGS, log
    torch.autograd.set_detect_anomaly(False)      # octree/optimization.py:131 switches it on at import
    purge()
    for name in ("svox", "svox.helpers", "absl", "absl.flags", "absl.app", "cv2", "imageio", "lpips"):
        if name in saved:
            sys.modules[name] = saved[name]
        else:
            sys.modules.pop(name, None)


class _Nerf:
    """Stand-in for octree.nerf.models.NerfModel.eval_points_raw: a smooth analytic field (a ball of density)."""
    use_viewdirs = False

    @staticmethod
    def field(points):
        p = points.double()
        r = p.norm(dim=-1, keepdim=True)
        sigma = 40.0 * (0.75 - r)
        rgb = torch.cat([torch.sin(p * (1.0 + 0.3 * c)) for c in range(K)], -1)          # [n, 12]
        return rgb.float(), sigma.float()

    def eval_points_raw(self, points, viewdirs=None, cross_broadcast=False):
        assert viewdirs is None
        return self.field(points)


def _poses(n, seed):
    from plenoctree_amd.nerf_sh.nerf.datasets import pose_spherical
    rs = np.random.RandomState(seed)
    return np.stack([pose_spherical(rs.uniform(0, 360), rs.uniform(-10, 60), 4.0) for _ in range(n)]).astype(f32)


# ------------------------------------------------------------------------------------------------------------------
def test_reference_extraction_steps_and_compression(ref, tmp_path, monkeypatch):
    FLAGS, log = ref
    ref_ex = importlib.import_module("octree.extraction")          # the reference's file, unmodified
    import svox
    depth, center, radius = 2, [0.0, 0.1, 0.0], [1.3, 1.2, 1.4]
    reso = 2 ** (depth + 1)
    args = types.SimpleNamespace(init_grid_depth=depth, z_min=None, z_max=None, chunk=256, alpha_thresh=0.01,
                                 samples_per_cell=4, use_viewdirs=False, sh_deg=1, projection_samples=100)
    dataset = types.SimpleNamespace(w=W, h=H, focal=FX, size=3, camtoworlds=_poses(3, 5))
    nerf = _Nerf()

    def new_tree():
        return svox.N3Tree(N=2, data_dim=D, init_refine=0, init_reserve=500000, geom_resize_fact=1.0, depth_limit=depth,
                           radius=radius, center=center, data_format=f"SH{K}", extra_data=None, map_location="cpu")

    # the mask the reference will compute, restated independently (octree/extraction.py:294-331)
    ot = T.Tree(D, depth, center, radius)
    pts = T.grid_points(reso, ot.offset, ot.invradius)
    sig = _Nerf.field(torch.from_numpy(pts))[1][:, 0].numpy()
    for mode in ("sigma", "weight"):
        FLAGS.masking_mode, FLAGS.weight_thresh, FLAGS.renderer_step_size = mode, 1e-3, 1e-2
        FLAGS.config, FLAGS.spherify = "blender", False
        tree = new_tree()
        ref_ex.step1(args, tree, nerf, dataset)
        if mode == "sigma":
            mask = sig >= -np.log(1.0 - args.alpha_thresh) / (2.0 / reso)
        else:
            w = None
            for c in dataset.camtoworlds:
                w = T.grid_weight_render(sig.reshape(reso, reso, reso).astype(f32), c, W, H, FX, T.RenderOptions(1e-2),
                                         ot.offset, ot.invradius, weight=w)
            mask = w.reshape(-1) >= 1e-3
            assert ("grid_weight", 1) in log                       # one _C.grid_weight_render call per camera
        assert 0 < mask.sum() < mask.size
        want = T.build_from_mask(mask.reshape(reso, reso, reso), depth, D, center, radius)
        assert np.array_equal(tree.child.numpy(), want.child) and np.array_equal(tree.parent_depth.numpy(), want.parent_depth)
        assert tree.max_depth == depth and tree.n_internal == want.n_internal
        # one-pass build from the same mask (what our own driver uses) gives the same arrays as the reference's loop
        assert np.array_equal(tree.depths.numpy(), want.depths())
    # ---- step 2 on the weight-mode tree ----
    del log[:]
    ref_ex.step2(args, tree, nerf)
    leaves = tree._leaf_packed()
    deep = leaves[tree.depths == tree.max_depth]
    drawn = [(e[1], e[2]) for e in log if e[0] == "sample"]
    assert torch.equal(torch.cat([p for p, _ in drawn]), deep)     # every deepest leaf once, in leaf order, chunked
    assert max(p.numel() for p, _ in drawn) <= args.chunk // args.samples_per_cell
    flat = tree.data.data.view(-1, D)
    for packed, points in drawn:
        rgb, sigma = _Nerf.field(points.view(-1, 3))
        expect = torch.cat([rgb, sigma], -1).reshape(-1, args.samples_per_cell, D).mean(dim=1)
        assert torch.allclose(flat[packed], expect, rtol=1e-6, atol=1e-6)
        # the points lie inside their own leaf
        again = sys.modules["plenoctree_amd.octree_ops"].tree_query(tree.child, points.view(-1, 3), tree.offset, tree.invradius)
        assert torch.equal(again.view(-1, args.samples_per_cell), packed[:, None].expand(-1, args.samples_per_cell))
    shallow = leaves[tree.depths != tree.max_depth]
    assert float(flat[shallow].abs().max()) == 0.0
    # ---- finish (octree/extraction.py:503-509) ----
    had_negative = bool((flat[:, -1] < 0).any())
    tree[:, -1:].relu_()
    assert had_negative and float(tree.data.data[..., -1].min()) == 0.0
    tree.shrink_to_fit()
    assert "svox.N3Tree(N=2, data_dim=13" in repr(tree) and f"capacity:{tree.n_internal}/" in repr(tree)
    path = str(tmp_path / "tree.npz")
    tree.save(path, compress=False)
    back = svox.N3Tree.load(path, map_location="cpu")
    assert torch.equal(back.child, tree.child) and torch.equal(back.parent_depth, tree.parent_depth)
    assert torch.equal(back.data.data, tree.data.data.half().float())
    assert torch.equal(back.offset, tree.offset) and torch.equal(back.invradius, tree.invradius)
    # ---- the reference's compression.py reads our file (octree/compression.py:76-86: the key list) ----
    ref_comp = importlib.import_module("octree.compression")
    out_dir = str(tmp_path / "min")
    monkeypatch.setattr(sys, "argv", ["compression.py", path, "--noquant", "--out_dir", out_dir])
    ref_comp.main()
    small = svox.N3Tree.load(os.path.join(out_dir, "tree.npz"), map_location="cpu")     # parent_depth dropped: rebuilt
    assert torch.equal(small.child, tree.child) and torch.equal(small.parent_depth, tree.parent_depth)
    assert torch.equal(small.data.data, back.data.data)
    z = np.load(os.path.join(out_dir, "tree.npz"))
    assert sorted(z.files) == sorted(["data_dim", "child", "invradius3", "offset", "data", "data_format"])
    from plenoctree_amd.octree import compression as our_comp
    assert our_comp.main([path, "--noquant", "--out_dir", str(tmp_path / "min_ours")]) and not our_comp.main(
        [path, "--noquant", "--out_dir", str(tmp_path / "min_ours")])                       # second call: "skip"
    zo = np.load(str(tmp_path / "min_ours" / "tree.npz"))
    assert sorted(zo.files) == sorted(z.files) and all(np.array_equal(zo[k], z[k]) for k in z.files)
    # ---- ... and its default mode: median-cut quantisation through _C.quantize_median_cut (:88-136), two ways:
    # every basis function quantised, and the first one retained (--retain 1); 6 bits = 64 colours per basis function
    for extra, tag in (([], "q"), (["--retain", "1"], "qr")):
        qdir = str(tmp_path / tag)
        monkeypatch.setattr(sys, "argv", ["compression.py", path, "--bits", "6", "--sigma_thresh", "0.5", "--out_dir", qdir,
                                          "--overwrite"] + extra)
        ref_comp.main()
        zq = np.load(os.path.join(qdir, "tree.npz"))
        n, kq = tree.n_internal, K - (1 if extra else 0)
        assert zq["quant_colors"].shape == (kq, 64, 3) and zq["quant_colors"].dtype == np.float16
        assert zq["quant_map"].shape == (kq, n, 2, 2, 2) and zq["quant_map"].dtype == np.uint16 and int(zq["quant_map"].max()) < 64
        assert zq["sigma"].shape == (n, 2, 2, 2) and "data" not in zq.files
        assert ("data_retained" in zq.files) == bool(extra)
        # our loader undoes it: sigma exact below/above the threshold, colours to the quantisation error
        q = svox.N3Tree.load(os.path.join(qdir, "tree.npz"), map_location="cpu")
        assert torch.equal(q.child, tree.child) and q.data.shape == back.data.shape
        sig, keep = back.data.data[..., -1], back.data.data[..., -1] > 0.5
        assert torch.equal(q.data.data[..., -1], torch.where(keep, sig, torch.zeros_like(sig)))
        err = (q.data.data[..., :-1] - back.data.data[..., :-1])[keep]
        ref_spread = back.data.data[..., :-1][keep].std()
        assert float(err.abs().mean()) < 0.5 * float(ref_spread), (float(err.abs().mean()), float(ref_spread))
        # our own driver (plenoctree_amd.octree.compression, same flags) writes the same file
        from plenoctree_amd.octree import compression as our_comp
        odir = str(tmp_path / (tag + "_ours"))
        assert our_comp.main([path, "--bits", "6", "--sigma_thresh", "0.5", "--out_dir", odir, "--overwrite"] + extra)
        zo = np.load(os.path.join(odir, "tree.npz"))
        assert sorted(zo.files) == sorted(zq.files)
        for k in zq.files:
            assert zo[k].dtype == zq[k].dtype and np.array_equal(zo[k], zq[k]), k
        if extra:                                              # the retained basis function is kept to float16
            d0 = back.data.data[..., :-1].reshape(n, 2, 2, 2, 3, K)[..., 0][keep]
            assert torch.equal(q.data.data[..., :-1].reshape(n, 2, 2, 2, 3, K)[..., 0][keep], d0.half().float())


def _small_tree_file(path, seed=3):
    """A depth-2 SH4 tree with random contents, written with OUR N3Tree.save."""
    from plenoctree_amd.octree import svox
    mask = np.random.RandomState(seed).rand(8, 8, 8) < 0.3
    t = T.build_from_mask(mask, 2, D, [0.0, 0.0, 0.0], [1.3, 1.3, 1.3])
    rs = np.random.RandomState(seed + 1)
    t.data[:] = (rs.randn(*t.data.shape) * 0.5).astype(f32)
    t.data[..., -1] = (np.abs(rs.randn(*t.data.shape[:-1])) * 3).astype(f32)
    h = svox.N3Tree(N=2, data_dim=D, depth_limit=2, radius=1.3, center=[0.0, 0.0, 0.0], data_format=f"SH{K}")
    h.child, h.parent_depth = torch.from_numpy(t.child.copy()), torch.from_numpy(t.parent_depth.copy())
    h.data = torch.nn.Parameter(torch.from_numpy(t.data.copy()))
    h.level_nodes = np.bincount(t.parent_depth[:, 1]).tolist()
    h.save(path, compress=False)
    t.data = h.data.data.half().float().numpy()          # what a loader sees (float16 on disk)
    return t


def _images(t, c2ws, seed):
    """Ground truth = renders of a perturbed copy of the tree, so that fitting has something to do."""
    rs = np.random.RandomState(seed)
    data = torch.from_numpy(t.data).double() + 0.3 * torch.from_numpy(rs.randn(*t.data.shape))
    out = []
    for c in c2ws:
        ro, rd = _rays(c)
        out.append(T.render_rays_torch(t, data, ro, rd, rd, T.RenderOptions(1e-2)).reshape(H, W, 3).clamp(0, 1).float().numpy())
    return np.stack(out)


def test_reference_optimization_main(ref, tmp_path, monkeypatch):
    FLAGS, log = ref
    ref_opt = importlib.import_module("octree.optimization")      # the reference's file, unmodified
    src, dst = str(tmp_path / "tree.npz"), str(tmp_path / "tree_opt.npz")
    t = _small_tree_file(src)
    c2w = {"train": _poses(3, 11), "val": _poses(2, 12)}
    sets = {k: types.SimpleNamespace(focal=FX, camtoworlds=v, images=_images(t, v, 7).reshape(len(v), -1, 3), h=H, w=W)
            for k, v in c2w.items()}
    monkeypatch.setattr(sys.modules["octree.nerf.datasets"], "get_dataset", lambda stage, args: sets[stage])
    LR = 300.0
    FLAGS.__dict__.update(input=src, output=dst, config=FLAGS._config_path, sgd=True, lr=LR, sgd_momentum=0.0,
                          sgd_nesterov=False, num_epochs=2, val_interval=1, split_train=False, render_interval=0,
                          nosave=False, continue_on_decrease=True, renderer_step_size=1e-2)
    ref_opt.main([])
    assert os.path.exists(dst)                                      # validation PSNR improved -> best tree saved
    assert [e for e in log if e[0] == "render_bwd"] == [("render_bwd", True)] * 6      # 2 epochs x 3 images, forward reused
    # independent restatement of the loop (octree/optimization.py:212-230): per-image SGD on the float64 oracle gradient
    data = torch.from_numpy(t.data).double()
    opt = T.RenderOptions(1e-2)
    for _ in range(2):
        for c, gt in zip(c2w["train"], sets["train"].images.reshape(-1, H, W, 3)):
            d = data.clone().requires_grad_(True)
            ro, rd = _rays(c)
            im = T.render_rays_torch(t, d, ro, rd, rd, opt).reshape(H, W, 3)
            ((im.float().clamp(0.0, 1.0) - torch.from_numpy(gt)) ** 2).mean().backward()
            data = (data.float() - LR * d.grad.float()).double()
    from plenoctree_amd.octree import svox
    got = svox.N3Tree.load(dst, map_location="cpu")
    assert torch.equal(got.child, torch.from_numpy(t.child))
    assert float((got.data.data.double() - data).abs().max()) <= 2e-3 * float(data.abs().max())        # float16 file
    assert float((got.data.data.double() - torch.from_numpy(t.data).double()).abs().max()) > 1e-2      # and it did move


def test_reference_evaluation_main(ref, tmp_path, monkeypatch):
    FLAGS, log = ref
    ref_eval = importlib.import_module("octree.evaluation")       # the reference's file, unmodified
    src = str(tmp_path / "tree.npz")
    t = _small_tree_file(src, seed=8)
    c2w = _poses(2, 21)
    ds = types.SimpleNamespace(w=W, h=H, focal=FX, size=2, camtoworlds=c2w, images=_images(t, c2w, 9))
    monkeypatch.setattr(sys.modules["octree.nerf.datasets"], "get_dataset", lambda stage, args: ds)
    FLAGS.__dict__.update(input=src, config=FLAGS._config_path, write_vid=None, write_images=None, renderer_step_size=1e-2,
                          no_early_stop=False, spherify=False)
    results = []
    ref_utils = sys.modules["octree.nerf.utils"]
    real = ref_utils.eval_octree
    monkeypatch.setattr(ref_utils, "eval_octree", lambda *a, **k: results.append(real(*a, **k)) or results[-1])
    ref_eval.main([])
    avg_psnr, avg_ssim, avg_lpips, frames = results[0]
    assert ("render", 0.01) in log                                  # fast=True: svox's early-stopping preset
    psnrs = []
    for c, gt in zip(c2w, ds.images):
        ro, rd = _rays(c)
        im = T.render_rays_torch(t, torch.from_numpy(t.data).double(), ro, rd, rd, T.RenderOptions.for_renderer(1e-2, True))
        psnrs.append(-10.0 * np.log10(float(((im.reshape(H, W, 3).float().clamp(0, 1) - torch.from_numpy(gt)) ** 2).mean())))
    assert abs(avg_psnr - float(np.mean(psnrs))) < 1e-4
    assert 0.0 < avg_ssim <= 1.0 and avg_lpips == 0.0
