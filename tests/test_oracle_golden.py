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
