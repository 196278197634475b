"""Closed-form known answers for the oracle stages the reference cannot pin
(JAX path not importable)."""
import math

import numpy as np
import torch

from oracle import nerf_oracle as O


def test_sample_along_rays():
    o = torch.zeros(5, 3)
    d = torch.tensor([[0.0, 0.0, -2.0]]).repeat(5, 1)
    z, pts = O.sample_along_rays(o, d, 64, 2.0, 6.0, None)
    np.testing.assert_allclose(z[0].numpy(), np.linspace(2.0, 6.0, 64, dtype=np.float32), rtol=1e-6)
    np.testing.assert_allclose(pts[..., 2].numpy(), -2.0 * z.numpy(), rtol=1e-6)
    t = torch.rand(5, 64)
    zr, _ = O.sample_along_rays(o, d, 64, 2.0, 6.0, t)
    base = torch.linspace(2.0, 6.0, 64)
    mids = 0.5 * (base[1:] + base[:-1])
    lower = torch.cat([base[:1], mids]); upper = torch.cat([mids, base[-1:]])
    assert torch.all(zr >= lower) and torch.all(zr <= upper)
    assert torch.all(zr[:, 1:] >= zr[:, :-1])


def test_volumetric_rendering_empty_and_opaque():
    B, S = 3, 64
    z = torch.linspace(2, 6, S)[None].repeat(B, 1)
    d = torch.tensor([[0.0, 0.0, -1.0]]).repeat(B, 1)
    rgb = torch.rand(B, S, 3)
    c, disp, acc, w = O.volumetric_rendering(rgb, torch.zeros(B, S, 1), z, d, True)
    np.testing.assert_allclose(c.numpy(), 1.0); np.testing.assert_allclose(acc.numpy(), 0.0)
    np.testing.assert_allclose(disp.numpy(), 1e10)
    sigma = torch.zeros(B, S, 1); sigma[:, 10] = 1e9
    c, disp, acc, w = O.volumetric_rendering(rgb, sigma, z, d, True)
    np.testing.assert_allclose(c.numpy(), rgb[:, 10].numpy(), atol=1e-6)
    np.testing.assert_allclose(acc.numpy(), 1.0, atol=1e-6)
    np.testing.assert_allclose(disp.numpy(), 1.0 / z[:, 10].numpy(), rtol=1e-5)
    # last sample: dist = 1e10 => any sigma>0 is opaque there
    sigma = torch.zeros(B, S, 1); sigma[:, -1] = 1e-3
    c, disp, acc, w = O.volumetric_rendering(rgb, sigma, z, d, False)
    np.testing.assert_allclose(w[:, -1].numpy(), 1.0, atol=1e-6)


def test_piecewise_constant_pdf_onehot_and_zero():
    bins = torch.linspace(2, 6, 64)[None].repeat(2, 1)
    bins = 0.5 * (bins[:, 1:] + bins[:, :-1])          # 63 knots
    w = torch.zeros(2, 62); w[:, 17] = 1.0
    z = O.piecewise_constant_pdf(bins, w, 128, None)
    assert z.shape == (2, 128)
    assert torch.all(z >= bins[:, 17:18] - 1e-6) and torch.all(z <= bins[:, 18:19] + 1e-6)
    assert torch.all(z[:, 1:] >= z[:, :-1])
    # all-zero weights => eps padding => uniform over the 62 bins
    z = O.piecewise_constant_pdf(bins, torch.zeros(2, 62), 128, None)
    u = torch.linspace(0, 1 - float(np.finfo(np.float32).eps), 128)
    expect = bins[0, 0] + u * (bins[0, -1] - bins[0, 0])
    np.testing.assert_allclose(z[0].numpy(), expect.numpy(), rtol=1e-4)


def test_sample_pdf_sorted_192():
    B = 4
    o = torch.zeros(B, 3); d = torch.randn(B, 3)
    zc, _ = O.sample_along_rays(o, d, 64, 2.0, 6.0, torch.rand(B, 64))
    w = torch.rand(B, 64)
    zf, pts = O.sample_pdf(0.5 * (zc[:, 1:] + zc[:, :-1]), w[:, 1:-1], o, d, zc, 128,
                           torch.rand(B, 128))
    assert zf.shape == (B, 192) and pts.shape == (B, 192, 3)
    assert torch.all(zf[:, 1:] >= zf[:, :-1])
    assert zf.min() >= 2.0 - 1e-5 and zf.max() <= 6.0 + 1e-5


def test_adam_three_steps():
    p = torch.tensor([1.0, -2.0], dtype=torch.float64)
    m = torch.zeros_like(p); v = torch.zeros_like(p)
    grads = [torch.tensor([0.5, -1.0], dtype=torch.float64),
             torch.tensor([0.25, 2.0], dtype=torch.float64),
             torch.tensor([-1.0, 0.0], dtype=torch.float64)]
    pm, mm, vm = p.numpy().copy(), np.zeros(2), np.zeros(2)
    for t, g in enumerate(grads):
        p, m, v = O.adam_update(p, m, v, g, 1e-2, t)
        gn = g.numpy()
        mm = 0.9 * mm + 0.1 * gn; vm = 0.999 * vm + 0.001 * gn * gn
        pm = pm - 1e-2 * (mm / (1 - 0.9 ** (t + 1))) / (np.sqrt(vm / (1 - 0.999 ** (t + 1))) + 1e-8)
    np.testing.assert_allclose(p.numpy(), pm, rtol=1e-12)
    # first step moves each coordinate by ~lr*sign(g)
    p1, _, _ = O.adam_update(torch.tensor([1.0]), torch.zeros(1), torch.zeros(1),
                             torch.tensor([3.0]), 1e-2, 0)
    np.testing.assert_allclose(p1.numpy(), [0.99], rtol=1e-6)


def test_lr_and_psnr():
    assert math.isclose(O.learning_rate_decay(0, 5e-4, 5e-6, 1000), 5e-4, rel_tol=1e-12)
    assert math.isclose(O.learning_rate_decay(1000, 5e-4, 5e-6, 1000), 5e-6, rel_tol=1e-12)
    assert math.isclose(O.learning_rate_decay(500, 5e-4, 5e-6, 1000), 5e-5, rel_tol=1e-12)
    assert math.isclose(float(O.compute_psnr(torch.tensor(1e-3))), 30.0, rel_tol=1e-6)


def test_generate_rays_center_pixel():
    c2w = np.eye(4, dtype=np.float32)[None]
    c2w[0, :3, 3] = [1, 2, 3]
    rays = O.generate_rays(8, 6, 10.0, c2w)
    assert rays.origins.shape == (1, 6, 8, 3)
    np.testing.assert_allclose(rays.directions[0, 3, 4], [0, 0, -1])
    np.testing.assert_allclose(rays.directions[0, 0, 0], [-0.4, 0.3, -1])
    np.testing.assert_allclose(np.linalg.norm(rays.viewdirs, axis=-1), 1.0, rtol=1e-6)
    np.testing.assert_allclose(rays.origins[0, 2, 2], [1, 2, 3])


def _tiny_problem(dtype):
    cfg = O.Cfg(num_coarse_samples=8, num_fine_samples=8, sh_deg=1, net_width=16, net_depth=6,
                max_deg_point=2, sparsity_npoints=16)
    gen = torch.Generator().manual_seed(1)
    params = [O.init_mlp_params(cfg, gen, dtype), O.init_mlp_params(cfg, gen, dtype)]
    flat = O.flatten_params(params)
    flat = flat + 0.05 * torch.randn(flat.shape, generator=gen, dtype=dtype)
    B = 6
    o = torch.randn(B, 3, generator=gen, dtype=dtype) * 0.1
    d = torch.randn(B, 3, generator=gen, dtype=dtype)
    rays = O.Rays(o, d, d / d.norm(dim=-1, keepdim=True))
    px = torch.rand(B, 3, generator=gen, dtype=dtype)
    t_rand = torch.rand(B, 8, generator=gen, dtype=dtype)
    u = torch.rand(B, 8, generator=gen, dtype=dtype)
    sp = (torch.rand(16, 3, generator=gen, dtype=dtype) * 2 - 1) * 1.5
    return cfg, flat, rays, px, t_rand, u, sp


def _fd_check(cfg, flat, rays, px, t_rand, u, sp, idx):
    total, stats, grad = O.loss_and_grad(flat, rays, px, cfg, t_rand, u, sp)
    assert math.isclose(float(total), float(stats["loss"] + stats["loss_c"] + stats["loss_sp"]),
                        rel_tol=1e-12)
    for i in idx:
        h = 1e-6
        fp = flat.clone(); fp[i] += h
        fm = flat.clone(); fm[i] -= h
        lp, _ = O.loss_fn(O.unflatten_params(fp, cfg), rays, px, cfg, t_rand, u, sp)
        lm, _ = O.loss_fn(O.unflatten_params(fm, cfg), rays, px, cfg, t_rand, u, sp)
        fd = float(lp - lm) / (2 * h)
        assert abs(fd - float(grad[i])) <= 1e-7 + 1e-4 * abs(fd), (i, fd, float(grad[i]))


def test_loss_grad_finite_difference_f64():
    """Finite differences see through sample_pdf's stop_gradient (model_utils.py:286), so
    the fine-MLP half is checked on the full graph and the coarse MLP on a coarse-only one."""
    cfg, flat, rays, px, t_rand, u, sp = _tiny_problem(torch.float64)
    gen = torch.Generator().manual_seed(2)
    half = flat.numel() // 2
    idx = (half + torch.randint(0, half, (12,), generator=gen)).tolist()
    _fd_check(cfg, flat, rays, px, t_rand, u, sp, idx)
    cfg.num_fine_samples = 0   # eval_points_raw then uses MLP_0 (models.py:165-168)
    idx = torch.randint(0, half, (12,), generator=gen).tolist()
    _fd_check(cfg, flat, rays, px, t_rand, None, sp, idx)


def test_dp_invariance_of_mean_of_shard_means():
    """pmean semantics (train.py:117): mean over shards of per-shard-mean grads equals the
    full-batch grad when every shard sees the same sparsity points."""
    cfg, flat, rays, px, t_rand, u, sp = _tiny_problem(torch.float64)
    _, _, g_full = O.loss_and_grad(flat, rays, px, cfg, t_rand, u, sp)
    gs = []
    for s in range(2):
        sl = slice(3 * s, 3 * s + 3)
        r = O.Rays(rays.origins[sl], rays.directions[sl], rays.viewdirs[sl])
        gs.append(O.loss_and_grad(flat, r, px[sl], cfg, t_rand[sl], u[sl], sp)[2])
    np.testing.assert_allclose(((gs[0] + gs[1]) / 2).numpy(), g_full.numpy(), rtol=1e-9, atol=1e-14)


def test_chunked_oracle_helpers_match_the_oracle():
    """tests/_helpers.py evaluates loss_fn / render over ray chunks at the BASELINE sizes; here it is checked against
    the unchunked oracle (float64: identical up to summation order)."""
    from _helpers import (make_params, make_rays, oracle_loss_and_grad_chunked, oracle_loss_and_grad_given_z,
                          oracle_render_chunked)
    cfg = O.Cfg(sparsity_npoints=64, weight_decay_mult=0.3)
    flat = make_params(cfg, bias_scale=0.2).double()
    B = 12
    rays = O.Rays(*[x.double() for x in make_rays(B, 3)])
    gen = torch.Generator().manual_seed(1)
    px = torch.rand(B, 3, generator=gen, dtype=torch.float64)
    t_rand = torch.rand(B, 64, generator=gen, dtype=torch.float64)
    u = torch.rand(B, 128, generator=gen, dtype=torch.float64)
    sp = (torch.rand(64, 3, generator=gen, dtype=torch.float64) * 2 - 1) * 1.5
    _, st, g = O.loss_and_grad(flat, rays, px, cfg, t_rand, u, sp)
    st_c, g_c = oracle_loss_and_grad_chunked(flat, rays, px, cfg, t_rand, u, sp, torch.float64, chunk=5)
    for k in st:
        assert abs(float(st[k]) - st_c[k]) <= 1e-12 * max(1.0, abs(float(st[k]))), k
    assert float((g - g_c).norm() / g.norm()) < 1e-12
    with torch.no_grad():
        ref, aux = O.render(O.unflatten_params(flat, cfg), rays, cfg, t_rand, u, return_aux=True)
    # the loss with the sample positions injected (gradient arbiter of the GPU tests) is the same function
    l, lc, g_z = oracle_loss_and_grad_given_z(flat, rays, px, cfg, aux["z_c"], aux["z_f"], sp, torch.float64)
    assert abs(l - float(st["loss"])) < 1e-14 and abs(lc - float(st["loss_c"])) < 1e-14
    assert float((g - g_z).norm() / g.norm()) < 1e-13
    got = oracle_render_chunked(flat, rays, cfg, t_rand, u, torch.float64, chunk=5)
    for lvl in range(2):
        for j in range(3):
            assert torch.equal(ref[lvl][j], got[lvl][j])


def test_adam_against_an_independent_implementation_of_the_same_rule():
    """flax.optim.Adam (flax>=0.3.1, call sites nerf_sh/nerf/models.py:44, nerf_sh/train.py:119) cannot be imported here, so
    `adam_update` restates its published rule: m, v EMAs, both bias-corrected, eps added to sqrt(v_hat) (OUTSIDE the root).
    torch.optim.Adam implements the same rule (Kingma & Ba, Algorithm 1; eps outside the root, bias correction of both
    moments, no weight decay, no amsgrad) in independent code: 6 steps with a changing learning rate must agree to float64
    round-off.  Not a flax pin -- a second implementation of the published rule that the oracle did not write."""
    gen = torch.Generator().manual_seed(5)
    p0 = torch.randn(1000, generator=gen, dtype=torch.float64)
    ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=1.0, betas=(0.9, 0.999), eps=1e-8)
    p, m, v = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0)
    for step in range(6):
        g = torch.randn(1000, generator=gen, dtype=torch.float64) * 10.0 ** (-(step % 3))
        lr = O.learning_rate_decay(step, 5e-4, 5e-6, 10)
        for group in opt.param_groups:
            group["lr"] = lr
        ref.grad = g.clone()
        opt.step()
        p, m, v = O.adam_update(p, m, v, g, lr, step)
        np.testing.assert_allclose(p.numpy(), ref.detach().numpy(), rtol=1e-12, atol=1e-15)
