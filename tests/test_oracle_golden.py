"""Pin the oracle's posenc / MLP / eval_sh against vectors produced by the
reference's own modules (tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import nerf_oracle as O


def _params_from_npz(g, cfg):
    params = []
    for mi in range(2):
        mlp = []
        for li in range(cfg.net_depth + 2):
            mlp.append((torch.tensor(g[f"MLP_{mi}.Dense_{li}.kernel"]),
                        torch.tensor(g[f"MLP_{mi}.Dense_{li}.bias"])))
        params.append(mlp)
    return params


def test_posenc_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "posenc.npz"))
    enc = O.posenc(torch.tensor(g["x"]), 0, 10).numpy()
    assert enc.shape == (37, 63)
    np.testing.assert_array_equal(enc, g["enc"])
    # known answer: posenc(0) = [0,0,0, 0 x30, 1 x30]
    np.testing.assert_allclose(enc[0], np.r_[np.zeros(33), np.ones(30)], atol=1e-7)


@pytest.mark.parametrize("deg", [3, 4])
def test_eval_points_raw_golden(golden_dir, deg):
    K = (deg + 1) ** 2
    g = np.load(os.path.join(golden_dir, f"eval_points_sh{K}.npz"))
    cfg = O.Cfg(sh_deg=deg)
    params = _params_from_npz(g, cfg)
    shapes = O.layer_shapes(cfg)
    assert [tuple(w.shape) for w, _ in params[0]] == shapes
    n = sum(w.numel() + b.numel() for w, b in params[0])
    assert n == {3: 505649, 4: 512588}[deg]
    pts = torch.tensor(g["points"])
    rgb_f, sig_f = O.eval_points_raw(params, pts, cfg)
    rgb_c, sig_c = O.eval_points_raw(params, pts, cfg, coarse=True)
    # same torch CPU kernels as the reference's twin => bitwise up to matmul blocking
    np.testing.assert_allclose(rgb_f.numpy(), g["raw_rgb_fine"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(sig_f.numpy(), g["raw_sigma_fine"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(rgb_c.numpy(), g["raw_rgb_coarse"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(sig_c.numpy(), g["raw_sigma_coarse"], rtol=0, atol=2e-6)
    # flatten/unflatten round trip keeps the arena order
    flat = O.flatten_params(params)
    assert flat.numel() == 2 * n
    back = O.unflatten_params(flat, cfg)
    for a, b in zip(params[1], back[1]):
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


def test_eval_sh_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "eval_sh.npz"))
    dirs = torch.tensor(g["dirs"])
    for deg in range(5):
        res = O.eval_sh(deg, torch.tensor(g[f"sh_{deg}"]), dirs[:, None])
        np.testing.assert_allclose(res.numpy(), g[f"res_{deg}"], rtol=1e-5, atol=2e-6)
    # deg 0 known answer
    sh = torch.tensor(g["sh_0"])
    np.testing.assert_allclose(O.eval_sh(0, sh, dirs[:, None]).numpy(),
                               0.28209479177387814 * sh[..., 0].numpy(), rtol=1e-6)


def test_generate_rays_and_psnr_golden(golden_dir):
    """octree/nerf/utils.py:401-445 (generate_rays) and :310-319 (compute_psnr), run from the reference itself."""
    from plenoctree_amd.nerf_sh.nerf import utils
    g = np.load(os.path.join(golden_dir, "generate_rays.npz"))
    w, h, focal = int(g["w"]), int(g["h"]), float(g["focal"])
    for gen in (O.generate_rays, utils.generate_rays):          # the oracle and the product's host version
        rays = gen(w, h, focal, g["c2w"])
        np.testing.assert_allclose(rays.origins, g["origins"], rtol=0, atol=0)
        np.testing.assert_allclose(rays.directions, g["directions"], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(rays.viewdirs, g["viewdirs"], rtol=1e-6, atol=1e-6)
    for m, p in zip(g["mse"], g["psnr"]):
        assert float(O.compute_psnr(torch.tensor(m))) == pytest.approx(float(p), rel=1e-6)
        assert utils.compute_psnr(float(m)) == pytest.approx(float(p), rel=1e-6)
    # compute_ssim (octree/nerf/utils.py:322-398), batched and single-image
    ssim = utils.compute_ssim(torch.tensor(g["ssim_im0"]), torch.tensor(g["ssim_im1"]), max_val=1.0)
    np.testing.assert_allclose(ssim.numpy(), g["ssim"], rtol=1e-5)
    one = utils.compute_ssim(torch.tensor(g["ssim_im0"][1]), torch.tensor(g["ssim_im1"][1]))
    assert float(one) == pytest.approx(float(g["ssim"][1]), rel=1e-5) and 0.0 < float(one) < 1.0


def test_loaders_match_the_reference_loaders(golden_dir, tmp_path):
    """Our Blender / NSVF loaders on the same on-disk scenes the reference's loaders were run on
    (octree/nerf/datasets.py, factor 0, white background)."""
    import sys
    sys.path.insert(0, golden_dir)
    import golden_scenes
    from plenoctree_amd.nerf_sh.nerf import datasets, utils
    g = np.load(os.path.join(golden_dir, "loaders.npz"))
    roots = {"blender": golden_scenes.write_blender(str(tmp_path / "blender")),
             "nsvf": golden_scenes.write_nsvf(str(tmp_path / "nsvf"))}
    for kind, root in roots.items():
        for split in ("train", "test"):
            a = utils.define_flags().parse_args(["--train_dir", "x", "--data_dir", root, "--dataset", kind])
            a.factor, a.white_bkgd = 0, True
            ds = datasets.get_dataset(split, a, torch.device("cpu"), batch_size=4)
            h, w, focal = g[f"{kind}_{split}_hwf"]
            assert (ds.h, ds.w) == (int(h), int(w)) and ds.focal == pytest.approx(float(focal), rel=1e-6)
            np.testing.assert_allclose(ds.camtoworlds, g[f"{kind}_{split}_camtoworlds"], rtol=1e-6, atol=1e-7)
            imgs = ds.images.reshape(ds.size, ds.h, ds.w, 3).numpy()
            np.testing.assert_allclose(imgs, g[f"{kind}_{split}_images"], rtol=0, atol=1e-6)
        # factor = 2: cv2.INTER_AREA at an exact 2x down-scaling is the mean of 2x2 blocks; the fixture was produced by
        # the reference's loader with that definition standing in for cv2.resize (cv2 is not installed anywhere here)
        a = utils.define_flags().parse_args(["--train_dir", "x", "--data_dir", root, "--dataset", kind])
        a.factor, a.white_bkgd = 2, True
        ds = datasets.get_dataset("train", a, torch.device("cpu"), batch_size=4)
        h, w, focal = g[f"{kind}_train_f2_hwf"]
        assert (ds.h, ds.w) == (int(h), int(w)) and ds.focal == pytest.approx(float(focal), rel=1e-6)
        np.testing.assert_allclose(ds.images.reshape(ds.size, ds.h, ds.w, 3).numpy(), g[f"{kind}_train_f2_images"],
                                   rtol=0, atol=1e-6)
    # the reference's Blender loader raises for any other factor (nerf_sh/nerf/datasets.py:213-216)
    a = utils.define_flags().parse_args(["--train_dir", "x", "--data_dir", roots["blender"], "--dataset", "blender"])
    a.factor, a.white_bkgd = 4, True
    with pytest.raises(ValueError):
        datasets.get_dataset("train", a, torch.device("cpu"), batch_size=4)


def test_sampling_compositing_pdf_match_the_reference_function_bodies(golden_dir):
    """tests/golden/model_utils.npz holds the outputs of the reference's own nerf_sh/nerf/model_utils.py functions
    (cast_rays, sample_along_rays :104-142, posenc :145-173, volumetric_rendering :176-222,
    piecewise_constant_pdf :225-286, sample_pdf :289-314) executed with numpy standing in for jax.numpy and the
    random draws injected through the `key` argument.  float32 both sides; tolerance = a few ulps of the values
    (different but equivalent operation orders inside numpy and torch reductions)."""
    g = np.load(os.path.join(golden_dir, "model_utils.npz"))
    t = lambda k: torch.tensor(g[k])
    o, d = t("origins"), t("directions")
    for lindisp in (0, 1):
        for randomized in (0, 1):
            z, pts = O.sample_along_rays(o, d, 64, 2.0, 6.0, t("t_rand") if randomized else None, lindisp=bool(lindisp))
            np.testing.assert_allclose(z.numpy(), g[f"z_l{lindisp}_r{randomized}"], rtol=2e-6, atol=1e-6)
            np.testing.assert_allclose(pts.numpy(), g[f"pts_l{lindisp}_r{randomized}"], rtol=2e-6, atol=4e-6)
    np.testing.assert_allclose(O.posenc(t("posenc_x"), 0, 10).numpy(), g["posenc_enc"], rtol=0, atol=2e-6)
    z = t("z_l0_r1")
    for white in (0, 1):
        comp, disp, acc, w = O.volumetric_rendering(t("vr_rgb"), t("vr_sigma"), z, d, bool(white))
        np.testing.assert_allclose(w.numpy(), g[f"vr_weights_w{white}"], rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(comp.numpy(), g[f"vr_comp_w{white}"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(acc.numpy(), g[f"vr_acc_w{white}"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(disp.numpy(), g[f"vr_disp_w{white}"], rtol=2e-5, atol=1e-6)
    assert g["vr_acc_w1"][0] == 0.0 and g["vr_disp_w1"][0] == np.float32(1e10)      # empty ray: the guarded division
    assert abs(float(g["vr_acc_w1"][1]) - 1.0) < 1e-6                               # opaque surface
    bins, wts, u = t("pdf_bins"), t("pdf_weights"), t("pdf_u")
    for randomized in (0, 1):
        s = O.piecewise_constant_pdf(bins, wts, 128, u if randomized else None)
        want = g[f"pdf_samples_r{randomized}"]
        # inverse-CDF samples are ill-conditioned inside nearly empty bins (they move by a fraction of the bin per
        # ulp of the cdf): agree tightly almost everywhere, and to a small fraction of a bin in the rest
        err = np.abs(s.numpy() - want)
        assert np.mean(err > 1e-5) < 0.02 and err.max() < 2e-3, (np.mean(err > 1e-5), err.max())
        zs, ps = O.sample_pdf(bins, wts, o, d, z, 128, u if randomized else None)
        err = np.abs(zs.numpy() - g[f"sample_pdf_z_r{randomized}"])
        assert zs.shape == (7, 192) and np.mean(err > 1e-5) < 0.02 and err.max() < 2e-3
        assert bool((zs[:, 1:] >= zs[:, :-1]).all())
        err = np.abs(ps.numpy() - g[f"sample_pdf_pts_r{randomized}"])
        assert np.mean(err > 2e-5) < 0.02 and err.max() < 4e-3


def test_render_matches_the_reference_nerf_model_call(golden_dir):
    """tests/golden/nerf_model.npz: the reference's own NerfModel.__call__ (nerf_sh/nerf/models.py:216-348) with
    its MLP (model_utils.py:43-94), run through the numpy jax/flax shim of make_golden.py on the weights of
    eval_points_sh16.npz -- pins the composition (which weights feed sample_pdf, SH before sigmoid, white
    background, coarse + fine outputs) of the oracle's `render`."""
    g = np.load(os.path.join(golden_dir, "nerf_model.npz"))
    gw = np.load(os.path.join(golden_dir, "eval_points_sh16.npz"))
    cfg = O.Cfg()
    params = _params_from_npz(gw, cfg)
    rays = O.Rays(*[torch.tensor(g[k]) for k in ("origins", "directions", "viewdirs")])
    for r in (0, 1):
        out = O.render(params, rays, cfg, torch.tensor(g["t_rand"]) if r else None, torch.tensor(g["u"]) if r else None)
        assert len(out) == 2
        for lvl, (rgb, disp, acc) in zip(("coarse", "fine"), out):
            np.testing.assert_allclose(rgb.numpy(), g[f"rgb_{lvl}_r{r}"], rtol=0, atol=2e-5)
            np.testing.assert_allclose(acc.numpy(), g[f"acc_{lvl}_r{r}"], rtol=0, atol=2e-5)
            # disp = acc / depth is ill-conditioned on nearly empty rays (acc ~ 1e-3)
            np.testing.assert_allclose(disp.numpy(), g[f"disp_{lvl}_r{r}"], rtol=2e-3, atol=1e-6)
    assert float(np.abs(g["acc_coarse_r1"] - 0.5).max()) > 0.3           # the rays see both empty and opaque space


def test_loss_matches_the_reference_train_step(golden_dir):
    """tests/golden/train_loss.npz: Stats returned by the reference's own train_step / loss_fn
    (nerf_sh/train.py:51-121) run through the shim (value only; 500 sparsity points, weight_decay_mult 0.1 so
    that every term of the total is exercised)."""
    g = np.load(os.path.join(golden_dir, "train_loss.npz"))
    gw = np.load(os.path.join(golden_dir, "eval_points_sh16.npz"))
    cfg = O.Cfg(sparsity_npoints=int(g["sparsity_npoints"]), weight_decay_mult=float(g["weight_decay_mult"]))
    params = _params_from_npz(gw, cfg)
    rays = O.Rays(*[torch.tensor(g[k]) for k in ("origins", "directions", "viewdirs")])
    sp = torch.tensor(-1.5 + 3.0 * g["sp_u"])                   # random.uniform(key, minval=-r, maxval=r), train.py:79
    total, stats = O.loss_fn(params, rays, torch.tensor(g["pixels"]), cfg, torch.tensor(g["t_rand"]), torch.tensor(g["u"]), sp)
    for k in ("loss", "loss_c", "weight_l2", "psnr", "psnr_c"):
        assert float(stats[k]) == pytest.approx(float(g[k]), rel=2e-5), k
    assert float(stats["loss_sp"]) == pytest.approx(float(g["loss_sp"]), rel=5e-3, abs=1e-9)
    assert float(total) == pytest.approx(float(g["total"]), rel=2e-5)


def test_render_with_gaussian_noise_matches_the_reference(golden_dir):
    """tests/golden/nerf_model_noise.npz: the reference's NerfModel.__call__ with noise_std = 0.3 (add_gaussian_noise,
    model_utils.py:317-332, on raw sigma between the MLP and the relu, models.py:258-264,318-324), the normal draws
    injected through the keys.  randomized=False must ignore noise_std."""
    g = np.load(os.path.join(golden_dir, "nerf_model_noise.npz"))
    gw = np.load(os.path.join(golden_dir, "eval_points_sh16.npz"))
    cfg = O.Cfg(noise_std=float(g["noise_std"]))
    params = _params_from_npz(gw, cfg)
    rays = O.Rays(*[torch.tensor(g[k]) for k in ("origins", "directions", "viewdirs")])
    t = lambda k: torch.tensor(g[k])
    with torch.no_grad():
        noisy = O.render(params, rays, cfg, t("t_rand"), t("u"), noise_c=t("noise_c"), noise_f=t("noise_f"))
        det = O.render(params, rays, cfg)
        plain = O.render(params, rays, O.Cfg(), t("t_rand"), t("u"))
    for r, out in ((1, noisy), (0, det)):
        for lvl, (rgb, disp, acc) in zip(("coarse", "fine"), out):
            np.testing.assert_allclose(rgb.numpy(), g[f"rgb_{lvl}_r{r}"], rtol=0, atol=2e-5)
            np.testing.assert_allclose(acc.numpy(), g[f"acc_{lvl}_r{r}"], rtol=0, atol=2e-5)
    assert float((noisy[1][0] - plain[1][0]).abs().max()) > 1e-2          # the noise is not a no-op


def _train_grad_inputs(g, gw, dtype):
    """Inputs of tests/golden/train_grad.npz in `dtype` (weights of eval_points_sh16.npz, sigma-head biases + shift)."""
    cfg = O.Cfg(sparsity_npoints=int(g["sparsity_npoints"]), weight_decay_mult=float(g["weight_decay_mult"]))
    params = _params_from_npz(gw, cfg)
    flat = O.flatten_params(params)
    n = flat.numel() // 2
    for mi in range(2):                                   # Dense_8 bias = the float just before Dense_9's kernel + bias
        flat[(mi + 1) * n - 48 - 48 * 256 - 1] += float(g["sigma_bias_shift"])
    # the shift is applied in float32, as the fixture's generator did: the fine-pass gradient has a condition number of
    # ~1e4 with respect to the coarse density (sample positions x 2^9 encoding frequencies), so a bias that differs in
    # the 8th digit moves MLP_1's gradient by 1e-4 even in float64
    flat = flat.to(dtype)
    rays = O.Rays(*[torch.tensor(g[k]).to(dtype) for k in ("origins", "directions", "viewdirs")])
    sp = -1.5 + 3.0 * torch.tensor(g["sp_u"]).to(dtype)    # random.uniform(key, minval=-r, maxval=r), train.py:79
    return cfg, flat, rays, torch.tensor(g["pixels"]).to(dtype), torch.tensor(g["t_rand"]).to(dtype), torch.tensor(g["u"]).to(dtype), sp


def test_gradient_against_the_references_loss_fn_under_autograd(golden_dir):
    """tests/golden/train_grad.npz: reverse-mode AD (torch) through the REFERENCE'S OWN loss_fn body -- train.py:68-116
    with models.py / model_utils.py / sh.py imported from the reference and torch standing in for jax.numpy
    (tests/golden/make_golden_grad.py).  The oracle's jax.value_and_grad restatement must give the same gradient:
    float64 against float64 to the fixture's storage rounding (it is kept as float32: 6e-8 per element), and in
    float32 no further away than the reference's own float32 evaluation is (stored in the fixture)."""
    g = np.load(os.path.join(golden_dir, "train_grad.npz"))
    gw = np.load(os.path.join(golden_dir, "eval_points_sh16.npz"))
    want = torch.tensor(g["grad"]).double()
    cfg, flat, rays, px, t_rand, u, sp = _train_grad_inputs(g, gw, torch.float64)
    total, stats, grad = O.loss_and_grad(flat, rays, px, cfg, t_rand, u, sp)
    rel = float((grad - want).norm() / want.norm())
    assert rel < 2e-7, rel
    assert float(want.norm()) == pytest.approx(float(g["grad_norm_f64"]), rel=1e-6)
    # every leaf on its own (a wrong small leaf would hide in the global norm)
    off = 0
    for mi in range(2):
        for li, (fi, fo) in enumerate(O.layer_shapes(cfg)):
            for n in (fi * fo, fo):
                a, b = grad[off:off + n], want[off:off + n]
                assert float((a - b).norm()) <= 2e-7 * float(b.norm()) + 1e-12, (mi, li, n)
                assert float(b.norm()) > 0, (mi, li, n)          # the loss reaches every leaf
                off += n
    for k in ("loss", "loss_c", "loss_sp", "weight_l2", "psnr", "psnr_c"):
        assert float(stats[k]) == pytest.approx(float(g[k + "_f64"]), rel=1e-9), k
    cfg, flat, rays, px, t_rand, u, sp = _train_grad_inputs(g, gw, torch.float32)
    _, stats32, grad32 = O.loss_and_grad(flat, rays, px, cfg, t_rand, u, sp)
    rel32 = float((grad32.double() - want).norm() / want.norm())
    assert rel32 < 3 * float(g["grad_f32_vs_f64_rel_l2"]), (rel32, float(g["grad_f32_vs_f64_rel_l2"]))
    for k in ("loss", "loss_c", "weight_l2"):
        assert float(stats32[k]) == pytest.approx(float(g[k + "_f32"]), rel=2e-5), k


def test_host_helpers_match_the_reference_utils(golden_dir):
    """pose_spherical (:656-685), learning_rate_decay (:483-515) and generate_rays (:545-589) of the reference's
    nerf_sh/nerf/utils.py, run through the shim."""
    from plenoctree_amd.nerf_sh.nerf import utils
    g = np.load(os.path.join(golden_dir, "nerf_sh_utils.npz"))
    for (t, ph, r, up), want in zip(g["pose_args"], g["poses"]):
        np.testing.assert_allclose(utils.pose_spherical(t, ph, r, int(up)), want, rtol=0, atol=1e-6)
    for a, want in zip(g["lr_args"], g["lr"]):
        for fn in (utils.learning_rate_decay, O.learning_rate_decay):
            assert fn(int(a[0]), a[1], a[2], int(a[3]), int(a[4]), a[5]) == pytest.approx(float(want), rel=1e-6)
    gr = np.load(os.path.join(golden_dir, "generate_rays.npz"))
    rays = utils.generate_rays(9, 7, 12.5, gr["c2w"])
    np.testing.assert_allclose(rays.directions, g["rays_directions"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(rays.viewdirs, g["rays_viewdirs"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(rays.origins, g["rays_origins"], rtol=0, atol=0)


def test_sh_basis_against_the_references_second_sh_implementation(golden_dir):
    """octree/nerf/sh_proj.py:56-239 (EvalSH, the hard-coded real SH the reference uses to project view-dependent
    NeRFs) is an implementation independent of nerf_sh/nerf/sh.py's eval_sh; the two oracles' basis functions must
    equal it band by band, index l (l + 1) + m, signs included."""
    g = np.load(os.path.join(golden_dir, "sh_proj.npz"))
    dirs = torch.tensor(g["dirs"])                                   # float64 unit vectors
    assert g["basis"].shape == (64, 25)
    ours = O.sh_basis(4, dirs).numpy()
    np.testing.assert_allclose(ours, g["basis"], rtol=0, atol=2e-7)  # the constants of sh.py are float32 literals
    for deg in range(4):                                             # lower degrees are prefixes of the same list
        K = (deg + 1) ** 2
        np.testing.assert_allclose(O.sh_basis(deg, dirs).numpy(), g["basis"][:, :K], rtol=0, atol=2e-7)
    # eval_sh is the contraction with that basis
    sh = torch.randn(64, 3, 25, dtype=torch.float64, generator=torch.Generator().manual_seed(1))
    np.testing.assert_allclose(O.eval_sh(4, sh, dirs).numpy(), np.einsum("nck,nk->nc", sh.numpy(), g["basis"]),
                               rtol=0, atol=2e-6)
    from oracle import octree_oracle as T                            # the octree oracle's per-ray basis (float32)
    for i in range(0, 64, 7):
        np.testing.assert_allclose(T.sh_basis_np(25, g["dirs"][i]).astype(np.float64), g["basis"][i], rtol=0, atol=2e-6)
