"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on identical
seeded inputs, and against the golden vectors generated from the reference's own modules.

Tolerances (float32 path; the MFMA f32 GEMM is an exact fmaf chain in a permuted K order):
  per-stage tensors  rtol 2e-4 / atol 2e-5   (8 chained 256-wide layers)
  rendered colours   atol 2e-5,  |dPSNR| <= 1e-4 dB  (north_star)
"""
import os

import numpy as np
import pytest
import torch

from oracle import nerf_oracle as O

pytestmark = pytest.mark.gpu


from _helpers import (_gpu, _ops, _psnr, close, hip_sample_positions, make_params, make_rays,  # noqa: E402
                      oracle_loss_and_grad_given_z, pxo_cfg, split_mlp)


# ---------------------------------------------------------------------------------------
def test_library_loads_and_layout():
    ops = _ops(); _gpu()
    for deg, n_expect in ((3, 505649), (4, 512588)):
        leaves, n = ops.param_layout(ops.make_cfg(sh_deg=deg))
        assert n == n_expect
        shapes = O.layer_shapes(O.Cfg(sh_deg=deg))
        off = 0
        for l, (fi, fo) in enumerate(shapes):
            assert leaves[2 * l] == (l, 0, off, fi, fo)
            off += fi * fo
            assert leaves[2 * l + 1][:3] == (l, 1, off)
            off += fo


def test_posenc_golden_and_oracle(golden_dir):
    ops = _ops(); dev = _gpu()
    g = np.load(os.path.join(golden_dir, "posenc.npz"))
    enc = ops.posenc(torch.tensor(g["x"], device=dev))
    close("posenc/golden", enc, torch.tensor(g["enc"]), rtol=0, atol=2e-6)
    x = (torch.rand(1000, 3) * 2 - 1) * 6.0
    close("posenc/oracle", ops.posenc(x.to(dev)), O.posenc(x, 0, 10), rtol=0, atol=2e-6)


def _expected_pack(flat_mlp, cfg):
    """numpy restatement of pxo_common.h packed_index for the forward image (layers 0..7)."""
    mlp = O.unflatten_params(torch.cat([flat_mlp, flat_mlp]), cfg)[0]
    imgs = []
    for l in range(8):
        w = mlp[l][0].numpy()
        K = {0: 64, 5: 320}.get(l, 256)
        wp = np.zeros((K, 256), np.float32)
        wp[:w.shape[0]] = w
        img = np.zeros(K * 256, np.float32)
        k = np.arange(K)[:, None]; n = np.arange(256)[None, :]
        g, kk = k // 8, k % 8
        lane = (kk // 4) * 32 + n % 32
        idx = ((g * 8 + n // 32) * 64 + lane) * 4 + kk % 4
        img[idx.reshape(-1)] = wp.reshape(-1)
        imgs.append(img)
    return np.concatenate(imgs)


def test_pack_weights_image():
    ops = _ops(); dev = _gpu()
    cfg = O.Cfg()
    flat = make_params(cfg)
    mlp0 = split_mlp(flat, cfg, 0)
    pf, pb = ops.pack_weights(pxo_cfg(ops, cfg), mlp0.to(dev))
    exp = _expected_pack(mlp0, cfg)
    got = pf.cpu().numpy()[:exp.size]
    np.testing.assert_array_equal(got, exp)
    # biases sit after the head image
    nf, nb = ops.packed_sizes(pxo_cfg(ops, cfg))
    assert pf.numel() == nf and pb.numel() == nb
    mlp = O.unflatten_params(flat, cfg)[0]
    bias_off = exp.size + 32 * 2 * 256
    for l in range(8):
        np.testing.assert_array_equal(pf.cpu().numpy()[bias_off + l * 256: bias_off + (l + 1) * 256], mlp[l][1].numpy())


@pytest.mark.parametrize("deg", [3, 4])
def test_eval_points_golden(golden_dir, deg):
    """HIP eval_points_raw against outputs of the reference's own torch NerfModel."""
    ops = _ops(); dev = _gpu()
    K = (deg + 1) ** 2
    g = np.load(os.path.join(golden_dir, f"eval_points_sh{K}.npz"))
    cfg = O.Cfg(sh_deg=deg)
    params = [[(torch.tensor(g[f"MLP_{mi}.Dense_{li}.kernel"]), torch.tensor(g[f"MLP_{mi}.Dense_{li}.bias"]))
               for li in range(10)] for mi in range(2)]
    flat = O.flatten_params(params)
    pcfg = pxo_cfg(ops, cfg)
    pts = torch.tensor(g["points"], device=dev)
    for which, tag in ((1, "fine"), (0, "coarse")):
        pf, _ = ops.pack_weights(pcfg, split_mlp(flat, cfg, which).to(dev), need_bwd=False)
        rgb, sigma = ops.eval_points(pcfg, pf, pts)
        close(f"eval_points/{tag}/rgb", rgb, torch.tensor(g[f"raw_rgb_{tag}"]))
        close(f"eval_points/{tag}/sigma", sigma, torch.tensor(g[f"raw_sigma_{tag}"]))
        _, sigma_only = ops.eval_points(pcfg, pf, pts, want_rgb=False)
        assert torch.equal(sigma_only, sigma)


@pytest.mark.parametrize("deg,M", [(3, 128 * 5 + 17), (4, 300), (1, 64), (3, 0)])
def test_mlp_fwd_saved_tensors(deg, M):
    ops = _ops(); dev = _gpu()
    cfg = O.Cfg(sh_deg=deg)
    flat = make_params(cfg)
    pcfg = pxo_cfg(ops, cfg)
    pf, _ = ops.pack_weights(pcfg, split_mlp(flat, cfg, 1).to(dev))
    pts = (torch.rand(M, 3, generator=torch.Generator().manual_seed(11)) * 2 - 1) * 3.0
    raw_rgb, raw_sigma, (acts, enc, mask) = ops.mlp_fwd(pcfg, pf, pts.to(dev), save=True)
    if M == 0:
        return
    mlp = O.unflatten_params(flat, cfg)[1]
    e = O.posenc(pts, 0, 10)
    rr, rs, ref_acts = O.mlp_forward(mlp, e, cfg, return_acts=True)
    close("enc", enc[:, :63], e, rtol=0, atol=2e-6)
    assert float(enc[:, 63].abs().max()) == 0.0
    for l in range(8):
        close(f"acts[{l}]", acts[l], ref_acts[l])
    close("raw_rgb", raw_rgb, rr)
    close("raw_sigma", raw_sigma, rs[:, 0])
    # inference variant gives identical results
    r2, s2 = ops.mlp_fwd(pcfg, pf, pts.to(dev), save=False)
    assert torch.equal(r2, raw_rgb) and torch.equal(s2, raw_sigma)


def _mlp_with_preacts(mlp, x, cfg):
    pre = []
    inputs = x
    for i in range(cfg.net_depth):
        z = x @ mlp[i][0] + mlp[i][1]
        z.retain_grad()
        pre.append(z)
        x = torch.relu(z)
        if i % cfg.skip_layer == 0 and i > 0:
            x = torch.cat([x, inputs], dim=-1)
    rs = x @ mlp[8][0] + mlp[8][1]
    rr = x @ mlp[9][0] + mlp[9][1]
    return rr, rs, pre


def _mlp_with_preacts_nograd(mlp, x, cfg):
    pre = []
    inputs = x
    for i in range(cfg.net_depth):
        z = x @ mlp[i][0] + mlp[i][1]
        pre.append(z)
        x = torch.relu(z)
        if i % cfg.skip_layer == 0 and i > 0:
            x = torch.cat([x, inputs], dim=-1)
    return None, None, pre


@pytest.mark.parametrize("deg,M", [(3, 128 * 3 + 40), (4, 200), (3, 9000)])
def test_mlp_backward(deg, M):
    ops = _ops(); dev = _gpu()
    cfg = O.Cfg(sh_deg=deg)
    flat = make_params(cfg)
    pcfg = pxo_cfg(ops, cfg)
    mlp_flat = split_mlp(flat, cfg, 1)
    pf, pb = ops.pack_weights(pcfg, mlp_flat.to(dev))
    gen = torch.Generator().manual_seed(13)
    pts = (torch.rand(M, 3, generator=gen) * 2 - 1) * 2.0
    C = cfg.num_rgb_channels
    d_rgb = torch.randn(M, C, generator=gen) * 0.1
    d_sigma = torch.randn(M, generator=gen) * 0.1
    # A pre-activation within float32 round-off of 0 may take either branch of the ReLU in two
    # correct float32 evaluations; such rows get a zero upstream gradient in BOTH paths so the
    # comparison below stays tight.
    with torch.no_grad():
        mlp_ng = O.unflatten_params(torch.cat([mlp_flat, mlp_flat]), cfg)[0]
        _, _, pre_ng = _mlp_with_preacts_nograd(mlp_ng, O.posenc(pts, 0, 10), cfg)
        ambiguous = torch.stack([(p.abs() < 1e-5).any(dim=1) for p in pre_ng]).any(dim=0)
    d_rgb[ambiguous] = 0.0
    d_sigma[ambiguous] = 0.0
    raw_rgb, raw_sigma, (acts, enc, mask) = ops.mlp_fwd(pcfg, pf, pts.to(dev), save=True)
    dz, dbias = ops.mlp_bwd_data(pcfg, pb, d_rgb.to(dev), d_sigma.to(dev), mask)
    grads = ops.mlp_bwd_weights(pcfg, acts, enc, dz, d_rgb.to(dev), d_sigma.to(dev), dbias)
    # oracle: autograd through the restated MLP
    leaf = mlp_flat.clone().requires_grad_(True)
    mlp = O.unflatten_params(torch.cat([leaf, leaf]), cfg)[0]
    rr, rs, pre = _mlp_with_preacts(mlp, O.posenc(pts, 0, 10), cfg)
    ((rr * d_rgb).sum() + (rs[:, 0] * d_sigma).sum()).backward()
    for l in range(8):
        close(f"dz[{l}]", dz[l], pre[l].grad, rtol=5e-4, atol=2e-6)
    gscale = float(leaf.grad.abs().max())
    close("param grads", grads, leaf.grad, rtol=1e-3, atol=1e-5 * max(gscale, 1.0))


@pytest.mark.parametrize("deg", [3, 4])
def test_half_height_tail_tiles(deg):
    """Launches larger than one round of tiles whose remainder fits one round at half height run that remainder as
    64-row tiles (pxo_common.h TileSched; e.g. the fine pass of BASELINE configs[1]: 24 rounds + 157 half tiles).
    Rows are independent of the tile they sit in, so every output row must be BIT-identical to the same row evaluated
    inside a small launch (full tiles only); the weight gradients, sums over rows, agree to summation order."""
    ops = _ops(); dev = _gpu()
    from plenoctree_amd import _lib
    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    n_head = cus * _lib.load().pxo_tile_rows()            # one whole round of full tiles
    N = n_head + 5000                                      # + 79 half tiles (40 full tiles' worth of rows)
    cfg = O.Cfg(sh_deg=deg); pcfg = pxo_cfg(ops, cfg)
    flat = make_params(cfg)
    mlp_flat = split_mlp(flat, cfg, 1).to(dev)
    pf, pb = ops.pack_weights(pcfg, mlp_flat)
    gen = torch.Generator().manual_seed(17)
    pts = ((torch.rand(N, 3, generator=gen) * 2 - 1) * 2.0).to(dev)
    C = cfg.num_rgb_channels
    d_rgb = (torch.randn(N, C, generator=gen) * 0.1).to(dev)
    d_sigma = (torch.randn(N, generator=gen) * 0.1).to(dev)
    parts = [slice(0, n_head), slice(n_head, n_head + 2048), slice(n_head + 2048, N)]     # each: full tiles only
    # inference path, with and without raw_rgb (sigma-only head)
    rgb, sig = ops.eval_points(pcfg, pf, pts)
    _, sig_only = ops.eval_points(pcfg, pf, pts, want_rgb=False)
    assert torch.equal(sig, sig_only)
    for sl in parts:
        r, s_ = ops.eval_points(pcfg, pf, pts[sl].contiguous())
        assert torch.equal(r, rgb[sl]) and torch.equal(s_, sig[sl])
    # training path: saved activations, relu mask -> backward(data) -> weight gradients
    raw_rgb, raw_sigma, (acts, enc, mask) = ops.mlp_fwd(pcfg, pf, pts, save=True)
    assert torch.equal(raw_rgb, rgb) and torch.equal(raw_sigma, sig[:, 0])
    dz, dbias = ops.mlp_bwd_data(pcfg, pb, d_rgb, d_sigma, mask)
    grads = ops.mlp_bwd_weights(pcfg, acts, enc, dz, d_rgb, d_sigma, dbias)
    g_sum = torch.zeros_like(grads, dtype=torch.float64)
    for sl in parts:
        p = pts[sl].contiguous(); dr = d_rgb[sl].contiguous(); ds = d_sigma[sl].contiguous()
        _, _, (a2, e2, m2) = ops.mlp_fwd(pcfg, pf, p, save=True)
        assert torch.equal(a2, acts[:, sl]) and torch.equal(e2, enc[sl])
        dz2, db2 = ops.mlp_bwd_data(pcfg, pb, dr, ds, m2)
        assert torch.equal(dz2, dz[:, sl])
        g_sum += ops.mlp_bwd_weights(pcfg, a2, e2, dz2, dr, ds, db2).double()
    gs = float(g_sum.abs().max())
    close("weight gradients: one launch vs sum of three", grads, g_sum, rtol=1e-4, atol=1e-6 * gs)


def test_sample_along_rays():
    ops = _ops(); dev = _gpu()
    rays = make_rays(300)
    t = torch.rand(300, 64, generator=torch.Generator().manual_seed(1))
    for t_rand in (None, t):
        z, pts = ops.sample_along_rays(rays.origins.to(dev), rays.directions.to(dev), 64, 2.0, 6.0,
                                       None if t_rand is None else t_rand.to(dev))
        zr, pr = O.sample_along_rays(rays.origins, rays.directions, 64, 2.0, 6.0, t_rand)
        close("z_vals", z, zr, rtol=0, atol=2e-6)
        close("points", pts, pr, rtol=0, atol=1e-5)
    z, _ = ops.sample_along_rays(rays.origins.to(dev), rays.directions.to(dev), 64, 0.5, 4.0, t.to(dev), lindisp=True)
    zr, _ = O.sample_along_rays(rays.origins, rays.directions, 64, 0.5, 4.0, t, lindisp=True)
    close("z_vals/lindisp", z, zr, rtol=1e-5, atol=2e-6)


def _composite_inputs(B, S, C, seed, opaque=False):
    gen = torch.Generator().manual_seed(seed)
    rays = make_rays(B, seed)
    raw_rgb = torch.randn(B, S, C, generator=gen)
    raw_sigma = torch.randn(B, S, 1, generator=gen) * (30.0 if opaque else 3.0)
    z, _ = O.sample_along_rays(rays.origins, rays.directions, S, 2.0, 6.0, torch.rand(B, S, generator=gen))
    return rays, raw_rgb, raw_sigma, z


def _oracle_composite(cfg, rays, raw_rgb, raw_sigma, z):
    K = cfg.sh_dim
    rgb = torch.sigmoid(O.eval_sh(cfg.sh_deg, raw_rgb.reshape(*raw_rgb.shape[:-1], 3, K), rays.viewdirs[:, None]))
    return O.volumetric_rendering(rgb, torch.relu(raw_sigma), z, rays.directions, cfg.white_bkgd)


@pytest.mark.parametrize("deg,S,white,opaque", [(3, 64, True, False), (3, 192, True, False), (4, 192, True, True),
                                                  (1, 40, False, False), (0, 7, True, False), (2, 130, True, True)])
def test_shade_composite_fwd_bwd(deg, S, white, opaque):
    ops = _ops(); dev = _gpu()
    cfg = O.Cfg(sh_deg=deg, white_bkgd=white)
    pcfg = pxo_cfg(ops, cfg)
    B = 37
    rays, raw_rgb, raw_sigma, z = _composite_inputs(B, S, cfg.num_rgb_channels, 17 + S, opaque)
    args = (raw_rgb.reshape(B * S, -1).to(dev), raw_sigma.reshape(-1).to(dev), z.to(dev), rays.directions.to(dev),
            rays.viewdirs.to(dev))
    comp, disp, acc, w = ops.shade_composite_fwd(pcfg, *args)
    rr = raw_rgb.clone().requires_grad_(True)
    rs = raw_sigma.clone().requires_grad_(True)
    c_ref, d_ref, a_ref, w_ref = _oracle_composite(cfg, rays, rr, rs, z)
    close("comp_rgb", comp, c_ref, rtol=1e-5, atol=2e-6)
    close("acc", acc, a_ref, rtol=1e-5, atol=2e-6)
    close("weights", w, w_ref, rtol=1e-4, atol=2e-6)
    close("disp", disp, d_ref, rtol=1e-4, atol=1e-6)
    g = torch.randn(B, 3, generator=torch.Generator().manual_seed(2))
    (c_ref * g).sum().backward()
    d_rgb, d_sigma = ops.shade_composite_bwd(pcfg, *args, g.to(dev))
    close("d_raw_rgb", d_rgb, rr.grad.reshape(B * S, -1), rtol=1e-4, atol=1e-6)
    close("d_raw_sigma", d_sigma, rs.grad.reshape(-1), rtol=2e-4, atol=1e-6 * max(1.0, float(rs.grad.abs().max())))


@pytest.mark.parametrize("deg,S,white,n_sp", [(3, 64, True, 0), (3, 192, True, 1000), (4, 192, True, 257), (1, 40, False, 3)])
def test_shade_composite_train_fused(deg, S, white, n_sp):
    """The one-launch training form (compositing + pixel loss + reverse + sparsity rows) against the loss of
    nerf_sh/train.py:77-98 differentiated by autograd on the oracle, and against the separate fwd / bwd kernels."""
    ops = _ops(); dev = _gpu()
    cfg = O.Cfg(sh_deg=deg, white_bkgd=white, sparsity_length=0.07, sparsity_weight=2e-3)
    pcfg = pxo_cfg(ops, cfg)
    B = 41
    C = cfg.num_rgb_channels
    rays, raw_rgb, raw_sigma, z = _composite_inputs(B, S, C, 29 + S, False)
    gen = torch.Generator().manual_seed(5)
    px = torch.rand(B, 3, generator=gen)
    sp_sigma = torch.randn(n_sp, generator=gen) * 20
    all_rgb = torch.cat([raw_rgb.reshape(B * S, C), torch.randn(n_sp, C, generator=gen)])
    all_sigma = torch.cat([raw_sigma.reshape(-1), sp_sigma])
    out = ops.shade_composite_train(pcfg, all_rgb.to(dev), all_sigma.to(dev), z.to(dev), rays.directions.to(dev),
                                    rays.viewdirs.to(dev), px.to(dev), n_sp=n_sp)
    rr = raw_rgb.clone().requires_grad_(True)
    rs = raw_sigma.clone().requires_grad_(True)
    sps = sp_sigma.clone().requires_grad_(True)
    c_ref, _, _, w_ref = _oracle_composite(cfg, rays, rr, rs, z)
    loss = ((c_ref - px) ** 2).mean()                                                   # train.py:89
    if n_sp:
        loss = loss + cfg.sparsity_weight * (1.0 - torch.exp(-cfg.sparsity_length * torch.relu(sps)).mean())   # :81-83
    loss.backward()
    close("comp_rgb", out["comp_rgb"], c_ref, rtol=1e-5, atol=2e-6)
    close("weights", out["weights"], w_ref, rtol=1e-4, atol=2e-6)
    close("ray_sse", out["ray_sse"], ((c_ref - px) ** 2).sum(-1), rtol=1e-4, atol=1e-7)
    close("d_raw_rgb", out["d_raw_rgb"][:B * S], rr.grad.reshape(B * S, -1), rtol=1e-4, atol=1e-8)
    close("d_raw_sigma", out["d_raw_sigma"][:B * S], rs.grad.reshape(-1), rtol=2e-4,
          atol=1e-6 * max(1.0, float(rs.grad.abs().max())))
    if n_sp:
        close("sparsity d_raw_sigma", out["d_raw_sigma"][B * S:], sps.grad, rtol=1e-5, atol=1e-12)
        close("sparsity exp", out["sp_exp"][:n_sp], torch.exp(-cfg.sparsity_length * torch.relu(sp_sigma)), rtol=1e-6, atol=1e-7)
        assert bool((out["d_raw_rgb"][B * S:] == 0).all())
    # identical to the stage kernels fed the same loss gradient
    args = (all_rgb[:B * S].to(dev), all_sigma[:B * S].to(dev), z.to(dev), rays.directions.to(dev), rays.viewdirs.to(dev))
    comp, _, _, w = ops.shade_composite_fwd(pcfg, *args)
    # (the compiler contracts multiply-adds differently in the two kernels: equal to float32 round-off, not bitwise)
    close("fused vs staged comp_rgb", out["comp_rgb"], comp, rtol=1e-6, atol=1e-7)
    close("fused vs staged weights", out["weights"], w, rtol=1e-6, atol=1e-9)
    d_rgb, d_sigma = ops.shade_composite_bwd(pcfg, *args, (comp - px.to(dev)) * (2.0 / (3 * B)))
    close("fused vs staged d_raw_rgb", out["d_raw_rgb"][:B * S], d_rgb, rtol=1e-5, atol=1e-10)
    close("fused vs staged d_raw_sigma", out["d_raw_sigma"][:B * S], d_sigma, rtol=1e-5,
          atol=1e-6 * max(1.0, float(d_sigma.abs().max())))


@pytest.mark.parametrize("deg", [3, 4])
def test_adam_pack_step_equals_adam_then_pack(deg):
    """pxo_adam_pack_step == pxo_adam_step followed by pxo_pack_weights on both MLPs, bit for bit (parameters,
    moments and all four fragment-ordered images, zero padding included)."""
    ops = _ops(); dev = _gpu()
    cfg = O.Cfg(sh_deg=deg); pcfg = pxo_cfg(ops, cfg)
    gen = torch.Generator().manual_seed(77 + deg)
    flat = make_params(cfg, bias_scale=0.2)
    n = flat.numel() // 2
    pa, pb = flat.to(dev), flat.to(dev)
    ma = torch.rand(2 * n, generator=gen).to(dev) * 1e-3; mb = ma.clone()
    va = torch.rand(2 * n, generator=gen).to(dev) * 1e-6; vb = va.clone()
    packed = [ops.pack_weights(pcfg, split_mlp(pa, cfg, i)) for i in range(2)]
    for step in range(2):
        g = (torch.randn(2 * n, generator=gen) * 1e-2).to(dev)
        ops.adam_pack_step(pcfg, pa, ma, va, g, 3e-4, step, packed, grad_scale=0.5)
        ops.adam_step(pb, mb, vb, g, 3e-4, step, grad_scale=0.5)
    report = {name: (int((a != b).sum()), float((a - b).abs().max())) for name, a, b in
              (("params", pa, pb), ("m", ma, mb), ("v", va, vb))}
    assert all(n == 0 for n, _ in report.values()), f"adam_pack_step vs adam_step (count differing, max abs): {report}"
    assert not torch.equal(pa.cpu(), flat)
    for i in range(2):
        f_ref, b_ref = ops.pack_weights(pcfg, split_mlp(pb, cfg, i))
        for name, a, b in ((f"forward image of MLP_{i}", packed[i][0], f_ref), (f"backward image of MLP_{i}", packed[i][1], b_ref)):
            bad = (a != b).nonzero().reshape(-1)
            assert bad.numel() == 0, f"{name}: {bad.numel()} of {a.numel()} differ, first at {bad[:5].tolist()}"


def test_composite_known_answers():
    ops = _ops(); dev = _gpu()
    cfg = O.Cfg(); pcfg = pxo_cfg(ops, cfg)
    B, S, C = 5, 64, 48
    rays = make_rays(B)
    z = torch.linspace(2, 6, S)[None].repeat(B, 1)
    raw_rgb = torch.randn(B * S, C)
    comp, disp, acc, w = ops.shade_composite_fwd(pcfg, raw_rgb.to(dev), torch.full((B * S,), -1.0, device=dev),
                                                 z.to(dev), rays.directions.to(dev), rays.viewdirs.to(dev))
    assert torch.all(comp == 1.0) and torch.all(acc == 0.0) and torch.all(disp == 1e10)   # empty space, white bg


@pytest.mark.parametrize("Nc,Nf", [(64, 128), (16, 8), (64, 192)])
def test_sample_pdf(Nc, Nf):
    ops = _ops(); dev = _gpu()
    B = 101
    gen = torch.Generator().manual_seed(23)
    rays = make_rays(B)
    zc, _ = O.sample_along_rays(rays.origins, rays.directions, Nc, 2.0, 6.0, torch.rand(B, Nc, generator=gen))
    w = torch.rand(B, Nc, generator=gen) ** 4
    w[0] = 0.0                       # eps-padding path
    w[1] = 0.0; w[1, Nc // 3] = 1.0  # one-hot
    w[2, : Nc // 2] = 0.0            # plateau in the cdf
    u = torch.rand(B, Nf, generator=gen)
    u[3, 0] = 0.0
    bins = 0.5 * (zc[:, 1:] + zc[:, :-1])
    for uu in (u, None):
        z, pts = ops.sample_pdf(zc.to(dev), w.to(dev), rays.origins.to(dev), rays.directions.to(dev), Nf,
                                None if uu is None else uu.to(dev))
        zr, pr = O.sample_pdf(bins, w[:, 1:-1], rays.origins, rays.directions, zc, Nf, uu)
        z = z.cpu()
        assert bool((z[:, 1:] >= z[:, :-1]).all()), "z not sorted"
        assert z.shape == (B, Nc + Nf)
        close("pts_fine", pts, rays.origins[:, None] + z[..., None] * rays.directions[:, None], rtol=0, atol=5e-6)
        # (1) inverse-CDF invariant in float64: F(z_fine) == u.  The fine samples are recovered from the
        # sorted union by removing the coarse ones (multiset difference).
        uu_ = uu if uu is not None else torch.linspace(0.0, 1.0 - float(np.finfo(np.float32).eps), Nf).expand(B, Nf)
        wd = w[:, 1:-1].double()
        wsum = wd.sum(-1, keepdim=True); pad = torch.clamp(1e-5 - wsum, min=0)
        pdf = (wd + pad / wd.shape[-1]) / (wsum + pad)
        cdf = torch.cat([torch.zeros(B, 1, dtype=torch.float64), torch.cumsum(pdf, -1)], -1)
        cdf[:, -1] = 1.0
        n_bad = 0
        for b in range(B):
            zf = z[b].tolist()
            for c in zc[b].tolist():
                zf.remove(c)                    # coarse values are copied bit-exactly
            zf = torch.tensor(sorted(zf), dtype=torch.float64)
            F = torch.from_numpy(np.interp(zf.numpy(), bins[b].double().numpy(), cdf[b].numpy()))
            us = torch.sort(uu_[b].double())[0]
            # float32 knots: cdf accurate to ~nbins*eps, z to ~eps*|z| mapped through the local pdf slope
            slope = (pdf[b] / (bins[b, 1:] - bins[b, :-1]).double()).max()
            tol = 64 * 1.2e-7 + float(slope) * 6.0 * 1.2e-7 * 4
            n_bad += int(((F - us).abs() > tol).sum())
        assert n_bad == 0, f"{n_bad} fine samples violate F(z) = u"
        # (2) element-wise against the oracle, with the tolerance scaled by the conditioning of the
        # inverse CDF (a sample inside a bin of mass dm moves by width*eps/dm per ulp of the cdf)
        bad = (z - zr).abs() > 1e-5
        if bad.any():
            cdf32 = cdf.float()
            for b, e in bad.nonzero().tolist():
                k = int(torch.searchsorted(bins[b].contiguous(), zr[b, e].clamp(bins[b, 0], bins[b, -1] - 1e-6), right=True)) - 1
                k = min(max(k, 0), Nc - 3)
                dm = float(cdf[b, k + 1] - cdf[b, k]); width = float(bins[b, k + 1] - bins[b, k])
                lim = 1e-5 + width * 16 * 1.2e-7 / max(dm, 1e-12)
                assert abs(float(z[b, e] - zr[b, e])) <= min(lim, width * 1.001), \
                    f"z_fine[{b},{e}] = {float(z[b, e])} vs oracle {float(zr[b, e])}, bin mass {dm:.3g}"


def test_uniform():
    ops = _ops(); _gpu()
    a = ops.uniform(1234, 0, 100003)
    assert float(a.min()) >= 0.0 and float(a.max()) < 1.0
    assert abs(float(a.mean()) - 0.5) < 5e-3 and abs(float(a.var()) - 1 / 12) < 5e-3
    assert torch.equal(a, ops.uniform(1234, 0, 100003))
    assert not torch.equal(a, ops.uniform(1234, 1, 100003))
    assert not torch.equal(a, ops.uniform(1235, 0, 100003))
    b = ops.uniform(7, 2, 999, -1.5, 1.5)
    assert float(b.min()) >= -1.5 and float(b.max()) < 1.5


def test_philox_generators_known_answers():
    """pxo_uniform / pxo_randint are Philox4x32-10 with counter = (block index, stream id) and key = seed: bit-for-bit the
    host restatement, which itself reproduces the Random123 known-answer vectors (tests/test_host_cpu.py) - including the
    first published vector (all-zero counter and key) as elements 0..3 of uniform(seed 0, stream 0)."""
    from _helpers import philox_randint, philox_uniform
    ops = _ops(); dev = _gpu()
    kat = np.array([0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8], np.uint64)
    got = ops.uniform(0, 0, 4, device=dev).cpu().numpy()
    assert np.array_equal(got, ((kat >> np.uint64(8)).astype(np.float32) / np.float32(16777216.0)))
    for seed, stream, n, lo, hi in ((0, 0, 4, 0.0, 1.0), (0x299F31D0A4093822, 0x0370734413198A2E, 1003, 0.0, 1.0),
                                    (20200823, 7, 257, -1.5, 1.5), (2 ** 64 - 1, 2 ** 64 - 1, 9, 2.0, 6.0)):
        got = ops.uniform(seed, stream, n, lo, hi, device=dev).cpu().numpy()
        want = philox_uniform(seed, stream, n, lo, hi)
        if (lo, hi) == (0.0, 1.0):
            assert np.array_equal(got, want), (seed, stream, n)                    # the 24-bit draw itself: exact
        else:                                                                        # lo + (hi - lo) r is one fma on the device
            assert np.abs(got - want).max() <= 1.5e-7 * (hi - lo) and got.min() >= lo and got.max() < hi, (seed, stream, n)
    for seed, stream, count, n in ((0, 0, 2, 2 ** 40), (5, 3, 1001, 640000), (2 ** 63 + 11, 2 ** 40 + 1, 7, 10)):
        got = ops.randint(seed, stream, count, n, device=dev).cpu().numpy()
        assert np.array_equal(got, philox_randint(seed, stream, count, n)), (seed, stream, count, n)


def test_adam_step():
    ops = _ops(); dev = _gpu()
    gen = torch.Generator().manual_seed(31)
    n = 100001
    p = torch.randn(n, generator=gen); m = torch.zeros(n); v = torch.zeros(n)
    pd, md, vd = p.to(dev), m.to(dev), v.to(dev)
    for step in range(3):
        g = torch.randn(n, generator=gen) * 10 ** (-step)
        lr = O.learning_rate_decay(step, 5e-4, 5e-6, 1000)
        ops.adam_step(pd, md, vd, (2 * g).to(dev), lr, step, grad_scale=0.5)
        p, m, v = O.adam_update(p, m, v, g, lr, step)
    close("adam/p", pd, p, rtol=1e-6, atol=1e-7)
    close("adam/m", md, m, rtol=1e-6, atol=1e-9)
    close("adam/v", vd, v, rtol=1e-6, atol=1e-12)
    # and against torch.optim.Adam: independent code for the same published rule (eps outside the root, both moments
    # bias-corrected), float64, the same gradients and learning rates
    gen = torch.Generator().manual_seed(31)
    ref = torch.randn(n, generator=gen).double().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=1.0, betas=(0.9, 0.999), eps=1e-8)
    for step in range(3):
        g = torch.randn(n, generator=gen) * 10 ** (-step)
        for group in opt.param_groups:
            group["lr"] = O.learning_rate_decay(step, 5e-4, 5e-6, 1000)
        ref.grad = g.double()
        opt.step()
    close("adam/p vs torch.optim.Adam", pd, ref.detach(), rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("deg,randomized", [(3, True), (3, False), (4, True)])
def test_render_fwd_matches_oracle(deg, randomized):
    ops = _ops(); dev = _gpu()
    cfg = O.Cfg(sh_deg=deg)
    if deg == 4:
        cfg.near, cfg.far = 0.0, 4.0
    pcfg = pxo_cfg(ops, cfg)
    flat = make_params(cfg, bias_scale=0.2)
    B = 200
    rays = make_rays(B, 41)
    gen = torch.Generator().manual_seed(43)
    t_rand = torch.rand(B, 64, generator=gen) if randomized else None
    u = torch.rand(B, 128, generator=gen) if randomized else None
    pk = [ops.pack_weights(pcfg, split_mlp(flat, cfg, i).to(dev), need_bwd=False)[0] for i in range(2)]
    out = ops.render_fwd(pcfg, pk[0], pk[1], rays.origins.to(dev), rays.directions.to(dev), rays.viewdirs.to(dev),
                         randomized=randomized, t_rand=None if t_rand is None else t_rand.to(dev),
                         u=None if u is None else u.to(dev))
    d64 = lambda t: None if t is None else t.double()
    with torch.no_grad():
        ref = O.render(O.unflatten_params(flat, cfg), rays, cfg, t_rand, u)
        ref64 = O.render(O.unflatten_params(flat.double(), cfg), O.Rays(*[x.double() for x in rays]), cfg,
                         d64(t_rand), d64(u))
    # The float64 oracle arbitrates: hierarchical resampling is ill-conditioned for samples that
    # fall into nearly empty bins (tests/test_gpu_parity.py::test_sample_pdf), so two correct float32
    # evaluations differ by up to ~1e-3 in a few pixels.  The HIP path must be as close to float64
    # as the float32 CPU oracle is, and agree with it to 1e-4 dB in PSNR (north_star).
    for lvl, tag in ((0, "coarse"), (1, "fine")):
        for j, name in ((0, "rgb"), (2, "acc")):
            got, r32, r64 = out[lvl][j].cpu().double(), ref[lvl][j].double(), ref64[lvl][j]
            assert torch.isfinite(got).all()
            e_hip, e_cpu = float((got - r64).abs().max()), float((r32 - r64).abs().max())
            assert e_hip <= 3 * e_cpu + 3e-5, f"{tag}/{name}: max err vs f64 {e_hip:.3g} (CPU f32: {e_cpu:.3g})"
            m_hip, m_cpu = float((got - r64).abs().mean()), float((r32 - r64).abs().mean())
            assert m_hip <= 3 * m_cpu + 2e-6, f"{tag}/{name}: mean err vs f64 {m_hip:.3g} (CPU f32: {m_cpu:.3g})"
        solid = ref64[lvl][2] > 0.05        # disp = acc/depth is ill-conditioned for nearly empty rays
        got, r32, r64 = out[lvl][1].cpu().double()[solid], ref[lvl][1].double()[solid], ref64[lvl][1][solid]
        e_hip, e_cpu = float(((got - r64) / r64).abs().max()), float(((r32 - r64) / r64).abs().max())
        # depth is the quantity most exposed to the ill-conditioned tail samples (u -> 1 lands in bins of
        # ~zero mass): bound it loosely, colour and acc above are the tight checks
        assert e_hip <= 3 * e_cpu + 5e-3, f"{tag}/disp: rel err vs f64 {e_hip:.3g} (CPU f32: {e_cpu:.3g})"
    close("coarse/rgb (elementwise, well-conditioned stage)", out[0][0], ref[0][0], rtol=0, atol=3e-5)
    # PSNR parity against an arbitrary target image: |dPSNR| <= 1e-4 dB (north_star)
    target = torch.rand(B, 3, generator=gen)
    assert abs(_psnr(out[1][0].cpu(), target) - _psnr(ref[1][0], target)) <= 1e-4
    assert abs(_psnr(out[1][0].cpu(), target) - _psnr(ref64[1][0], target)) <= 1e-4


@pytest.mark.parametrize("deg,Nf,sp", [(3, 128, True), (4, 128, True), (3, 0, True), (3, 128, False)])
def test_train_fwd_bwd_matches_oracle(deg, Nf, sp):
    ops = _ops(); dev = _gpu()
    cfg = O.Cfg(sh_deg=deg, num_fine_samples=Nf, sparsity_npoints=300, sparsity_weight=1e-3 if sp else 0.0)
    pcfg = pxo_cfg(ops, cfg)
    flat = make_params(cfg, bias_scale=0.2)
    B = 48
    rays = make_rays(B, 51)
    gen = torch.Generator().manual_seed(53)
    px = torch.rand(B, 3, generator=gen)
    t_rand = torch.rand(B, 64, generator=gen)
    u = torch.rand(B, max(Nf, 1), generator=gen)
    sp_pts = (torch.rand(300, 3, generator=gen) * 2 - 1) * 1.5
    fd = flat.to(dev)
    packed = [ops.pack_weights(pcfg, split_mlp(fd, cfg, i)) for i in range(2)]
    grads = torch.full_like(fd, float("nan"))
    stats = torch.zeros(6, device=dev)
    ws = torch.empty(ops.train_workspace_bytes(pcfg, B), dtype=torch.uint8, device=dev)
    ops.train_fwd_bwd(pcfg, fd, packed, rays.origins.to(dev), rays.directions.to(dev), rays.viewdirs.to(dev),
                      px.to(dev), grads, stats, ws, randomized=True, t_rand=t_rand.to(dev), u=u.to(dev),
                      sp_points=sp_pts.to(dev))
    total, st, g_ref = O.loss_and_grad(flat, rays, px, cfg, t_rand, u if Nf > 0 else None, sp_pts)
    s = stats.cpu()
    for i, k in enumerate(("loss", "psnr", "loss_c", "loss_sp", "psnr_c", "weight_l2")):
        close(f"stats/{k}", s[i], st[k].float(), rtol=2e-5, atol=1e-6)
    # Gradient.  End to end (positions drawn by each path) the comparison is dominated by the inverse-CDF step: a
    # fine sample that falls into a nearly empty bin moves by a fraction of a bin per ulp of the cdf, so at 48 rays
    # two correct float32 evaluations can be several % away from float64 (that step has its own test: test_sample_pdf,
    # F(z) = u, and the rendered colours).  The arbiter used here is therefore the float64 loss_fn evaluated AT THE
    # HIP PATH'S OWN SAMPLE POSITIONS (no gradient flows through them, model_utils.py:286): everything downstream --
    # posenc, both MLPs forward and backward, SH, compositing, the three loss terms -- is held to a fixed bound.
    # What remains is float32 round-off and ReLU-kink branch differences (~1e-4 each).
    rays_dev = (rays.origins.to(dev), rays.directions.to(dev), rays.viewdirs.to(dev))
    z_c, z_f = hip_sample_positions(ops, pcfg, packed, rays_dev, t_rand.to(dev), u.to(dev))
    l64, lc64, g64 = oracle_loss_and_grad_given_z(flat, rays, px, cfg, z_c.cpu(), None if z_f is None else z_f.cpu(),
                                                  sp_pts)
    close("loss vs f64 at the same positions", s[0], torch.tensor(l64), rtol=1e-5, atol=1e-7)
    if Nf > 0:
        close("loss_c vs f64 at the same positions", s[2], torch.tensor(lc64), rtol=1e-5, atol=1e-7)
    n = flat.numel() // 2
    halves = [(0, n)] + ([(n, 2 * n)] if Nf > 0 else [])
    for lo, hi in halves:
        ref = g64[lo:hi]
        assert float(ref.norm()) > 1e-4, "degenerate test: oracle gradient vanishes"
        e_hip = float((grads[lo:hi].cpu().double() - ref).norm() / ref.norm())
        assert e_hip <= 1e-3, f"MLP_{lo // n}: rel L2 err vs the f64 oracle at the same sample positions {e_hip:.3g}"
    if Nf == 0:
        assert float(grads[n:].abs().max()) == 0.0
    # end to end (each path draws its own positions): HIP float32 against the CPU float32 oracle
    e_f32 = float((grads.cpu().double() - g_ref.double()).norm() / g_ref.double().norm())
    assert e_f32 <= 2e-2, f"rel L2 err vs the float32 oracle, end to end: {e_f32:.3g}"


def test_grid_sigma_matches_eval_points():
    ops = _ops(); dev = _gpu()
    cfg = O.Cfg(); pcfg = pxo_cfg(ops, cfg)
    flat = make_params(cfg, bias_scale=0.2)
    pf, _ = ops.pack_weights(pcfg, split_mlp(flat, cfg, 1).to(dev), need_bwd=False)
    reso, radius, center = 12, torch.tensor([1.4, 1.5, 1.3]), torch.tensor([0.1, 0.0, -0.2])
    scale = 0.5 / radius; offset = 0.5 * (1.0 - center / radius)      # octree/extraction.py:250-251
    arr = (torch.arange(reso, dtype=torch.float32) + 0.5) / reso
    xx, yy, zz = [(arr - offset[i]) / scale[i] for i in range(3)]
    grid = torch.stack(torch.meshgrid(xx, yy, zz, indexing="ij")).reshape(3, -1).T.contiguous()
    sig = ops.grid_sigma(pcfg, pf, reso, 0, reso, offset.tolist(), scale.tolist())
    _, ref = O.eval_points_raw(O.unflatten_params(flat, cfg), grid, cfg)
    close("grid sigma", sig, ref[:, 0])
    # x-slab sharding (extraction voxel-sharded across GPUs): slabs concatenate to the full grid
    parts = [ops.grid_sigma(pcfg, pf, reso, x0, x0 + 4, offset.tolist(), scale.tolist()) for x0 in (0, 4, 8)]
    assert torch.equal(torch.cat(parts), sig)


# ---------------------------------------------------------------------------------------
# full-size (BASELINE.json configs[1]) size-independent properties: determinism, data-parallel invariance.
# (Oracle comparisons at this size live in tests/test_gpu_fullsize.py.)
# ---------------------------------------------------------------------------------------
def test_full_size_properties():
    ops = _ops(); dev = _gpu()
    cfg = O.Cfg(); pcfg = pxo_cfg(ops, cfg)
    flat = make_params(cfg, bias_scale=0.2).to(dev)
    B = 4096
    rays = make_rays(B, 61)
    o, d, v = rays.origins.to(dev), rays.directions.to(dev), rays.viewdirs.to(dev)
    px = torch.rand(B, 3, device=dev)
    t_rand = ops.uniform(1, 0, B * 64).reshape(B, 64)
    u = ops.uniform(1, 1, B * 128).reshape(B, 128)
    sp = ops.uniform(1, 2, 30000, -1.5, 1.5).reshape(10000, 3)
    packed = [ops.pack_weights(pcfg, split_mlp(flat, cfg, i)) for i in range(2)]

    def grads_for(sl):
        n = sl.stop - sl.start
        g = torch.full_like(flat, float("nan")); st = torch.zeros(6, device=dev)
        ws = torch.empty(ops.train_workspace_bytes(pcfg, n), dtype=torch.uint8, device=dev)
        ops.train_fwd_bwd(pcfg, flat, packed, o[sl].contiguous(), d[sl].contiguous(), v[sl].contiguous(),
                          px[sl].contiguous(), g, st, ws, randomized=True, t_rand=t_rand[sl].contiguous(),
                          u=u[sl].contiguous(), sp_points=sp)
        torch.cuda.synchronize()
        return g, st

    g_full, st_full = grads_for(slice(0, B))
    assert torch.isfinite(g_full).all() and torch.isfinite(st_full).all()
    # determinism: the same step twice is bit-identical (fixed-order reductions, no float atomics)
    g_again, _ = grads_for(slice(0, B))
    assert torch.equal(g_full, g_again)
    # data-parallel invariance (pmean, train.py:117): mean of the two half-batch gradients == full batch
    g0, st0 = grads_for(slice(0, B // 2)); g1, st1 = grads_for(slice(B // 2, B))
    gm = 0.5 * (g0 + g1)
    rel = float((gm - g_full).double().norm() / g_full.double().norm())
    assert rel < 1e-5, f"DP invariance violated: rel {rel}"
    close("DP loss", 0.5 * (st0[0] + st1[0]), st_full[0], rtol=1e-5, atol=0)
    # render: fine z sorted inside [near, far], acc in [0,1]
    out = ops.render_fwd(pcfg, packed[0][0], packed[1][0], o, d, v, randomized=True, seed=9)
    acc = out[1][2]
    assert float(acc.min()) >= 0.0 and float(acc.max()) <= 1.0 + 1e-5
    assert torch.isfinite(out[1][0]).all()


def _check_gen_mesh(common, train_dir):
    """nerf_sh/gen_mesh.py on the restored checkpoint: the sigma grid comes from the HIP evaluator, the OBJ holds the
    isosurface of exactly that grid, and the surface really sits on the level set (sigma re-evaluated at the vertices)."""
    from plenoctree_amd.nerf_sh import gen_mesh
    from plenoctree_amd.nerf_sh.nerf import models, utils
    dev = torch.device("cuda:0")
    args = gen_mesh.define_flags().parse_args(common)
    utils.update_flags(args)
    model, state = models.get_model_state(args, dev, restore=True)
    fn = lambda p: model.eval_points_raw(state, p, want_rgb=False)[1]
    # a 4 cm box: at the scene scale the posenc'd field (frequencies up to 2^9) is noise on a 40^3 grid, and "sigma at the
    # vertices == iso" only holds where the grid resolves the field
    reso, c1, c2 = [40, 36, 32], [-0.02] * 3, [0.02] * 3
    sig = gen_mesh.sigma_grid(fn, c1, c2, reso, 7001, dev)            # ragged chunks
    assert sig.shape == tuple(reso)
    # against the evaluator on the explicit meshgrid point list (the reference's construction, gen_mesh.py:105-111)
    grid = np.vstack(np.meshgrid(*(np.linspace(lo, hi, sz, dtype=np.float32) for lo, hi, sz in zip(c1, c2, reso)),
                                 indexing="ij")).reshape(3, -1).T
    ref = fn(torch.from_numpy(np.ascontiguousarray(grid)).to(dev)).reshape(*reso)
    assert torch.equal(sig, ref)
    iso = float(sig.median())
    verts, faces = gen_mesh.main(common + ["--reso", "40 36 32", "--c1", "-0.02", "--c2", "0.02", "--iso", repr(iso),
                                           "--point_chunk", "7001"])
    assert len(verts) > 100 and len(faces) > 100 and faces.min() >= 0 and faces.max() < len(verts)
    lines = open(os.path.join(train_dir, "mesh.obj")).read().splitlines()
    assert sum(l.startswith("v ") for l in lines) == len(verts) and sum(l.startswith("f ") for l in lines) == len(faces)
    # undo the reference's (c2-c1)/reso scaling (:127) to get back to sample space, then re-evaluate sigma there
    idx = (verts - np.array(c1)) * np.array(reso) / (np.array(c2) - np.array(c1))
    pos = np.array(c1) + idx * (np.array(c2) - np.array(c1)) / (np.array(reso) - 1)
    s_at = fn(torch.from_numpy(pos.astype(np.float32)).to(dev)).reshape(-1).cpu().numpy()
    spread = float(sig.std())
    assert np.median(np.abs(s_at - iso)) < 0.05 * spread, (np.median(np.abs(s_at - iso)), spread)


def test_cli_train_eval_extraction(tmp_path):
    """The drop-in entry points end to end on the synthetic scene: loss falls, checkpoint
    round-trips, eval renders with deterministic sampling, extraction evaluates the sigma grid."""
    _gpu()
    from plenoctree_amd.nerf_sh import train, eval as eval_mod
    from plenoctree_amd.octree import extraction
    # NB: like the reference (nerf_sh/nerf/utils.py:233-244) the YAML preset overrides command-line
    # flags for the keys it contains, so the short schedule has to live in the YAML itself.
    cfg_path = os.path.join(str(tmp_path), "short.yaml")
    with open(cfg_path, "w") as f:
        f.write("dataset: synthetic\nfactor: 4\nnum_coarse_samples: 64\nnum_fine_samples: 128\nuse_viewdirs: false\n"
                "white_bkgd: true\nbatch_size: 2048\nsh_deg: 3\nrandomized: true\nmax_steps: 60\nprint_every: 20\n"
                "save_every: 60\nrender_every: 0\n")
    common = ["--train_dir", str(tmp_path), "--config", cfg_path]
    trace = train.main(common)
    assert len(trace) == 3 and trace[-1][1] < trace[0][1], trace          # (step, loss, psnr, rays/s, avg_loss, avg_psnr)
    # avg_loss / avg_psnr = means over EVERY step since the last print (train.py:216-218), accumulated on the device: of the
    # size of the per-step values (batches of different images are noisy), falling from window to window, and consistent with
    # each other (mean psnr >= psnr of the mean loss, within a dB at this noise level)
    for (_, loss, _, _, avg_loss, avg_psnr) in trace:
        assert 0.5 * loss < avg_loss < 2.0 * loss and 0.0 <= avg_psnr + 10.0 * np.log10(avg_loss) < 1.0, trace
    assert trace[-1][4] < trace[0][4], trace
    assert os.path.exists(os.path.join(str(tmp_path), "checkpoint_60"))
    psnrs = eval_mod.main(common + ["--approx_eval_skip", "100", "--chunk", "4096", "--save_output", "false"])   # 2 images of 200x200
    assert len(psnrs) == 2 and all(np.isfinite(psnrs)) and min(psnrs) > 5.0
    # the same evaluation with the opt-in split-precision MLP forward: the same PSNR to well inside 1e-3 dB
    psnrs_x3 = eval_mod.main(common + ["--approx_eval_skip", "100", "--chunk", "4096", "--save_output", "true",
                                        "--mlp_precision", "bf16x3"])
    assert max(abs(a - b) for a, b in zip(psnrs, psnrs_x3)) < 1e-3, (psnrs, psnrs_x3)
    preds = os.path.join(str(tmp_path), "test_preds")                    # nerf_sh/eval.py:64-66,107-129
    for name in ("000.png", "disp_000.png", "100.png", "disp_100.png", "psnr.txt", "ssim.txt", "psnrs_60.txt", "ssims_60.txt"):
        assert os.path.exists(os.path.join(preds, name)), name
    assert abs(float(open(os.path.join(preds, "psnr.txt")).read()) - np.mean(psnrs_x3)) < 1e-6
    with pytest.raises(ValueError, match="inference option"):
        train.main(common + ["--mlp_precision", "bf16x3"])
    from plenoctree_amd.nerf_sh import gen_video
    frames = gen_video.main(common + ["--num_views", "2", "--height", "20", "--width", "24", "--chunk", "256",
                                      "--write_poses", os.path.join(str(tmp_path), "poses.txt")])
    assert len(frames) == 2 and frames[0].shape == (20, 24, 3) and frames[0].dtype == np.uint8
    assert os.path.exists(os.path.join(str(tmp_path), "video", "e300", "frames", "0001.png"))
    assert os.path.exists(os.path.join(str(tmp_path), "video", "e300", "video.gif"))
    assert np.loadtxt(os.path.join(str(tmp_path), "poses.txt")).shape == (8, 4)
    _check_gen_mesh(common, str(tmp_path))
    # the extraction driver restores the same checkpoint and evaluates its sigma grid (32^3 here); the complete
    # extraction -> optimisation -> evaluation chain is exercised in tests/test_gpu_octree.py
    from plenoctree_amd.nerf_sh.nerf import models, utils
    args = utils.define_flags().parse_args(common)
    utils.update_flags(args)
    model, state = models.get_model_state(args, torch.device("cuda:0"), restore=True)
    assert state.step > 0                                                  # restored, not freshly initialised
    sig = extraction.grid_sigma(model, state, 32, [0.0, 0.0, 0.0], [1.5, 1.5, 1.5])
    assert sig.shape == (32 ** 3,) and bool(torch.isfinite(sig).all())


def test_generate_rays_and_randint():
    ops = _ops(); dev = _gpu()
    from plenoctree_amd.nerf_sh.nerf import datasets
    W, H, focal = 37, 29, 41.5
    c2w = np.stack([datasets.pose_spherical(33.0, 21.0, 4.0311), datasets.pose_spherical(250.0, -7.0, 4.0311)])
    ref = O.generate_rays(W, H, focal, c2w)                      # [2,H,W,3] x3 (nerf_sh/nerf/utils.py:545-589)
    cd = torch.from_numpy(c2w).to(dev)
    for i in range(2):
        o, d, v = ops.generate_rays(cd[i], W, H, focal)          # whole image, pixel ids 0..H*W-1
        close("origins", o, torch.from_numpy(np.ascontiguousarray(ref.origins[i])).reshape(-1, 3), rtol=0, atol=0)
        close("directions", d, torch.from_numpy(ref.directions[i]).reshape(-1, 3), rtol=1e-6, atol=1e-6)
        close("viewdirs", v, torch.from_numpy(ref.viewdirs[i]).reshape(-1, 3), rtol=1e-6, atol=1e-6)
    ids = ops.randint(5, 3, 10001, W * H)
    assert ids.dtype == torch.int64 and int(ids.min()) >= 0 and int(ids.max()) < W * H
    assert torch.equal(ids, ops.randint(5, 3, 10001, W * H)) and not torch.equal(ids, ops.randint(6, 3, 10001, W * H))
    assert abs(float(ids.double().mean()) / (W * H) - 0.5) < 0.02
    o, d, v = ops.generate_rays(cd[1], W, H, focal, ids)
    pick = ids.cpu()
    close("directions[ids]", d, torch.from_numpy(ref.directions[1]).reshape(-1, 3)[pick], rtol=1e-6, atol=1e-6)
    # image_batching sampler (datasets.py:137-141,152-157): ids into the flattened [n_img, H*W] ray table
    gids = ops.randint(9, 1, 5003, 2 * W * H)
    assert int(gids.max()) >= W * H
    o, d, v = ops.generate_rays_multi(cd, W, H, focal, gids)
    pick = gids.cpu()
    for got, want in ((o, ref.origins), (d, ref.directions), (v, ref.viewdirs)):
        close("multi-camera rays", got, torch.from_numpy(np.ascontiguousarray(want)).reshape(-1, 3)[pick], rtol=1e-6, atol=1e-6)


def test_image_batching_feeder_on_device():
    """--image_batching true (nerf_sh/nerf/datasets.py:137-141,152-157): one batch draws rays from all images; every
    (ray, pixel) pair of the batch is the pair of its own image, and a train step runs on it."""
    ops = _ops(); dev = _gpu()
    from plenoctree_amd.nerf_sh.nerf import datasets, models, utils
    args = utils.define_flags().parse_args(["--config", "synthetic", "--train_dir", "x"])
    utils.update_flags(args); args.factor = 16; args.image_batching = True
    ds = datasets.get_dataset("train", args, dev, batch_size=512)
    b = next(ds)
    hw = ds.h * ds.w
    assert b["pixels"].shape == (512, 3) and b["rays"].origins.shape == (512, 3)
    cams = torch.from_numpy(ds.camtoworlds[:, :3, 3]).to(dev)
    cam_of_ray = (b["rays"].origins[:, None, :] - cams[None]).abs().sum(-1).argmin(dim=1)
    assert len(torch.unique(cam_of_ray)) > 20                                  # spans many images
    for c in torch.unique(cam_of_ray)[:5].tolist():
        full = ds.get_image(c)
        sel = cam_of_ray == c
        d_all = full["rays"].directions.reshape(-1, 3); px_all = full["pixels"].reshape(-1, 3)
        idx = (b["rays"].directions[sel][:, None, :] - d_all[None]).abs().sum(-1).argmin(dim=1)
        assert float((d_all[idx] - b["rays"].directions[sel]).abs().max()) < 1e-6
        assert torch.equal(px_all[idx], b["pixels"][sel])
    model, params = models.construct_nerf(args, dev)
    state = models.TrainState(model.cfg, params)
    stats = models.train_step(model, state, b, 5e-4, seed=1)
    assert bool(torch.isfinite(stats).all()) and state.step == 1


def test_mean_over_samples():
    ops = _ops(); dev = _gpu()
    for deg, S in ((3, 8), (4, 256)):
        cfg = O.Cfg(sh_deg=deg); pcfg = pxo_cfg(ops, cfg)
        C = cfg.num_rgb_channels
        gen = torch.Generator().manual_seed(S)
        rgb = torch.randn(37 * S, C, generator=gen); sigma = torch.randn(37 * S, 1, generator=gen)
        out = ops.mean_over_samples(pcfg, rgb.to(dev), sigma.to(dev), S)
        ref = torch.cat([rgb, sigma], -1).reshape(-1, S, C + 1).mean(dim=1)      # octree/extraction.py:391-393
        close("leaf mean", out, ref, rtol=1e-5, atol=1e-6)


def test_trained_psnr_matches_oracle_training(golden_dir):
    """north_star: PSNR of a HIP-trained model within 0.1 dB of the oracle-trained one.  Both run the same 400 Adam
    steps (64 rays x (64+128) samples + 1000 sparsity points per step, the reference's log-linear lr schedule
    5e-4 -> 5e-6 annealed over the horizon, nerf_sh/nerf/utils.py:483-515) from the same initialisation with identical
    batches and injected randoms (tests/_helpers.py:twin_short_steps -- seeds only); the result is compared on held-out
    rays rendered with deterministic sampling.  Over this horizon the held-out PSNR rises by more than 5 dB
    (10.0 -> ~15.5 dB), so the 0.1 dB bar is a small fraction of the training signal.

    The oracle leg (~1.5 min of host CPU) was run once by `tests/golden/make_trained_twin.py short`; its two PSNRs are the
    fixture trained_twin_64x400.json.  PXO_TWIN_LIVE_ORACLE=1 (or a missing fixture) trains the oracle inside the test
    instead, step by step next to the HIP leg, as rounds 1-4 did (15.5214 dB on the GPU box, profiles/r05i_trained_psnr.json)."""
    ops = _ops(); dev = _gpu()
    import json
    from _helpers import TWIN_SHORT_RAYS, TWIN_SHORT_SPARSITY, TWIN_SHORT_STEPS, twin_heldout, twin_short_steps
    from plenoctree_amd.nerf_sh.nerf import models, utils
    cfg = O.Cfg(sparsity_npoints=TWIN_SHORT_SPARSITY)
    pcfg = pxo_cfg(ops, cfg)
    fixture = os.path.join(golden_dir, f"trained_twin_{TWIN_SHORT_RAYS}x{TWIN_SHORT_STEPS}.json")
    live = os.environ.get("PXO_TWIN_LIVE_ORACLE", "0") == "1" or not os.path.exists(fixture)
    if not live:
        # the committed leg is only as good as what it was computed from: the oracle's source and the inputs (seeds, sampler,
        # scene, initialisation) are hashed into the fixture; if either differs today the oracle is trained live instead
        from _helpers import twin_short_digests
        with open(fixture) as f:
            g0 = json.load(f)
        now = twin_short_digests(cfg)
        stale = [k for k in now if g0.get(k) != now[k]]
        if stale:
            print(f"trained_twin fixture is stale ({stale} changed): training the oracle leg live")
            live = True
    flat0 = O.flatten_params(O.init_params(cfg, seed=20200823))
    model = models.NerfModel(pcfg)
    state = models.TrainState(pcfg, flat0.clone().to(dev))
    p, m, v = flat0.clone(), torch.zeros_like(flat0), torch.zeros_like(flat0)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    for step, batch, t_rand, u, sp, lr in twin_short_steps(cfg):
        if live:
            p, m, v, _, _ = O.train_step(p, m, v, step, O.Rays(*batch["rays"]), batch["pixels"], cfg, t_rand, u, sp, lr)
        dbatch = {"rays": utils.Rays(*[r.to(dev) for r in batch["rays"]]), "pixels": batch["pixels"].to(dev)}
        models.train_step(model, state, dbatch, lr, t_rand=t_rand.to(dev), u=u.to(dev), sp_points=sp.to(dev))
    rays, px = twin_heldout()                                            # three held-out views, every 4th pixel
    with torch.no_grad():
        if live:
            psnr_ref = _psnr(O.render(O.unflatten_params(p, cfg), rays, cfg)[1][0], px)
            psnr_init = _psnr(O.render(O.unflatten_params(flat0, cfg), rays, cfg)[1][0], px)
        else:
            with open(fixture) as f:
                g = json.load(f)
            assert (g["rays_per_step"], g["steps"], g["sparsity_npoints"]) == (TWIN_SHORT_RAYS, TWIN_SHORT_STEPS, TWIN_SHORT_SPARSITY)
            psnr_ref, psnr_init = float(g["psnr_trained"]), float(g["psnr_init"])
        # HIP-trained weights through the float64 oracle: the arbiter of same-weights render parity
        rays64 = O.Rays(*[r.double() for r in rays])
        cross = O.render(O.unflatten_params(state.params.cpu().double(), cfg), rays64, cfg)[1][0]
    out = model.apply(state, utils.Rays(*[r.to(dev) for r in rays]), False)[1][0].cpu()
    psnr_hip, psnr_cross = _psnr(out, px), _psnr(cross, px)
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "trained_psnr.json"), "w") as f:
            f.write('{"steps": %d, "rays_per_step": %d, "oracle_leg": "%s", "psnr_init": %.4f, "psnr_oracle_trained": %.4f, '
                    '"psnr_hip_trained": %.4f, "psnr_hip_trained_oracle_rendered": %.4f}\n'
                    % (TWIN_SHORT_STEPS, TWIN_SHORT_RAYS, "live" if live else "fixture", psnr_init, psnr_ref, psnr_hip, psnr_cross))
    assert psnr_ref > psnr_init + 4.0, (psnr_ref, psnr_init)      # the horizon carries a real training signal
    assert abs(psnr_hip - psnr_ref) <= 0.1, (psnr_hip, psnr_ref)
    assert abs(psnr_hip - psnr_cross) <= 1e-4, (psnr_hip, psnr_cross)   # same TRAINED weights: render parity at north_star's bar


def test_trained_psnr_twin_512_rays(golden_dir):
    """The 0.1 dB bar at the per-GPU step of the reference's 4096-ray batch on 8 devices: 300 Adam steps of 512 rays x
    (64+128) samples + 10,000 sparsity points.  The oracle leg (float32, ~9 minutes of CPU) was run once by
    tests/golden/make_trained_twin.py and its final parameters are the fixture trained_twin_512x300.npz; this test replays
    the same batches and injected randoms (tests/_helpers.py:twin_steps -- seeds only) through the HIP path and compares
    held-out PSNRs (every 4th pixel of three test views, deterministic sampling):
      |PSNR(HIP-trained, HIP-rendered) - PSNR(oracle-trained, oracle-rendered)| <= 0.1 dB          (north_star)
      |PSNR(HIP-trained, HIP-rendered) - PSNR(HIP-trained, float64-oracle-rendered)| <= 1e-4 dB   (same weights)
    and the horizon must carry a training signal of more than 3 dB (9.98 -> 13.94 dB in the fixture)."""
    ops = _ops(); dev = _gpu()
    from _helpers import TWIN_RAYS, TWIN_STEPS, twin_heldout, twin_steps
    from plenoctree_amd.nerf_sh.nerf import models, utils
    g = np.load(os.path.join(golden_dir, f"trained_twin_{TWIN_RAYS}x{TWIN_STEPS}.npz"))
    assert int(g["rays_per_step"]) == TWIN_RAYS and int(g["steps"]) == TWIN_STEPS
    cfg = O.Cfg()
    pcfg = pxo_cfg(ops, cfg)
    flat0 = O.flatten_params(O.init_params(cfg, seed=20200823))
    model = models.NerfModel(pcfg)
    state = models.TrainState(pcfg, flat0.clone().to(dev))
    for step, batch, t_rand, u, sp, lr in twin_steps(TWIN_RAYS, TWIN_STEPS, cfg):
        dbatch = {"rays": utils.Rays(*[r.to(dev) for r in batch["rays"]]), "pixels": batch["pixels"].to(dev)}
        models.train_step(model, state, dbatch, lr, t_rand=t_rand.to(dev), u=u.to(dev), sp_points=sp.to(dev))
    rays, px = twin_heldout()
    out = model.apply(state, utils.Rays(*[r.to(dev) for r in rays]), False)[1][0].cpu()
    with torch.no_grad():
        ref = O.render(O.unflatten_params(torch.tensor(g["params"]), cfg), rays, cfg)[1][0]
        rays64 = O.Rays(*[r.double() for r in rays])
        cross = O.render(O.unflatten_params(state.params.cpu().double(), cfg), rays64, cfg)[1][0]
    psnr_hip, psnr_ref, psnr_cross = _psnr(out, px), _psnr(ref, px), _psnr(cross, px)
    print(f"twin {TWIN_RAYS} rays x {TWIN_STEPS} steps: init {float(g['psnr_init']):.3f} dB, oracle-trained {psnr_ref:.4f} dB "
          f"(fixture says {float(g['psnr_trained']):.4f}), HIP-trained {psnr_hip:.4f} dB, same weights through the f64 oracle "
          f"{psnr_cross:.6f} dB")
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "trained_psnr_twin512.json"), "w") as f:
            f.write('{"steps": %d, "rays_per_step": %d, "psnr_init": %.4f, "psnr_oracle_trained": %.4f, '
                    '"psnr_hip_trained": %.4f, "psnr_hip_trained_f64_oracle_rendered": %.6f}\n'
                    % (TWIN_STEPS, TWIN_RAYS, float(g["psnr_init"]), psnr_ref, psnr_hip, psnr_cross))
    assert psnr_ref == pytest.approx(float(g["psnr_trained"]), abs=2e-3)       # the fixture's weights render as recorded
    assert psnr_ref > float(g["psnr_init"]) + 3.0
    assert abs(psnr_hip - psnr_ref) <= 0.1, (psnr_hip, psnr_ref)
    assert abs(psnr_hip - psnr_cross) <= 1e-4, (psnr_hip, psnr_cross)


@pytest.mark.parametrize("N", [1, 63, 127, 128, 129, 1000])
def test_eval_points_ragged_sizes(N):
    """Tile tails: every size around the 64/128-row tile boundary gives the same rows as a padded batch."""
    ops = _ops(); dev = _gpu()
    cfg = O.Cfg(); pcfg = pxo_cfg(ops, cfg)
    flat = make_params(cfg)
    pf, _ = ops.pack_weights(pcfg, split_mlp(flat, cfg, 1).to(dev), need_bwd=False)
    pts = (torch.rand(1024, 3, generator=torch.Generator().manual_seed(N)) * 2 - 1) * 2.0
    full_rgb, full_sig = ops.eval_points(pcfg, pf, pts.to(dev))
    rgb, sig = ops.eval_points(pcfg, pf, pts[:N].contiguous().to(dev))
    assert rgb.shape == (N, 48) and sig.shape == (N, 1)
    assert torch.equal(rgb, full_rgb[:N]) and torch.equal(sig, full_sig[:N])     # row-independent, bit-exact


@pytest.mark.parametrize("deg,B", [(0, 5), (1, 3), (2, 1), (4, 2)])
def test_train_step_small_and_all_degrees(deg, B):
    """Every SH degree the reference supports (nerf_sh/nerf/sh.py:69) through the whole train step, with
    tiny / odd ray counts (partial tiles in every kernel)."""
    ops = _ops(); dev = _gpu()
    cfg = O.Cfg(sh_deg=deg, sparsity_npoints=50)
    pcfg = pxo_cfg(ops, cfg)
    flat = make_params(cfg, bias_scale=0.2)
    rays = make_rays(B, 71 + deg)
    gen = torch.Generator().manual_seed(73)
    px = torch.rand(B, 3, generator=gen)
    t_rand = torch.rand(B, 64, generator=gen); u = torch.rand(B, 128, generator=gen)
    sp = (torch.rand(50, 3, generator=gen) * 2 - 1) * 1.5
    fd = flat.to(dev)
    packed = [ops.pack_weights(pcfg, split_mlp(fd, cfg, i)) for i in range(2)]
    grads = torch.full_like(fd, float("nan")); stats = torch.zeros(6, device=dev)
    ws = torch.empty(ops.train_workspace_bytes(pcfg, B), dtype=torch.uint8, device=dev)
    ops.train_fwd_bwd(pcfg, fd, packed, rays.origins.to(dev), rays.directions.to(dev), rays.viewdirs.to(dev),
                      px.to(dev), grads, stats, ws, randomized=True, t_rand=t_rand.to(dev), u=u.to(dev),
                      sp_points=sp.to(dev))
    total, st, g_ref = O.loss_and_grad(flat, rays, px, cfg, t_rand, u, sp)
    for i, k in enumerate(("loss", "psnr", "loss_c", "loss_sp", "psnr_c", "weight_l2")):
        close(f"stats/{k}", stats.cpu()[i], st[k].float(), rtol=3e-4, atol=2e-6)   # few rays: no averaging of the
        # ill-conditioned fine samples (see test_sample_pdf)
    assert float(g_ref.norm()) > 0
    rel = float((grads.cpu().double() - g_ref.double()).norm() / g_ref.double().norm())
    assert rel < 2e-3, f"gradient relative L2 error {rel}"      # kink-flip bound, see the larger test


def test_error_paths_on_device():
    """C-ABI error behaviour: bad sizes / missing buffers return codes and messages, nothing launches."""
    ops = _ops(); dev = _gpu()
    from plenoctree_amd import _lib
    cfg = O.Cfg(); pcfg = pxo_cfg(ops, cfg)
    z = torch.zeros(4, 300, device=dev)
    with pytest.raises(_lib.PxoError, match="samples per ray"):
        ops.shade_composite_fwd(pcfg, torch.zeros(1200, 48, device=dev), torch.zeros(1200, device=dev), z,
                                torch.ones(4, 3, device=dev), torch.ones(4, 3, device=dev))
    with pytest.raises(_lib.PxoError, match="sample_pdf"):
        ops.sample_pdf(torch.zeros(4, 2, device=dev), torch.zeros(4, 2, device=dev), torch.zeros(4, 3, device=dev),
                       torch.ones(4, 3, device=dev), 8)
    with pytest.raises(_lib.PxoError, match="workspace"):
        flat = make_params(cfg).to(dev)
        packed = [ops.pack_weights(pcfg, split_mlp(flat, cfg, i)) for i in range(2)]
        ops.train_fwd_bwd(pcfg, flat, packed, torch.zeros(8, 3, device=dev), torch.ones(8, 3, device=dev),
                          torch.ones(8, 3, device=dev), torch.zeros(8, 3, device=dev), torch.zeros_like(flat),
                          torch.zeros(6, device=dev), torch.empty(1024, dtype=torch.uint8, device=dev))
    with pytest.raises(_lib.PxoError):
        ops.posenc(torch.zeros(4, 3))          # host tensor
    # round-4 entry points: negative noise, an unknown profiling tag, a workspace too small for the work report,
    # a configuration field out of range
    with pytest.raises(_lib.PxoError, match="pxo_add_gaussian_noise"):
        ops.add_gaussian_noise(torch.zeros(8, device=dev), -1.0)
    with pytest.raises(_lib.PxoError, match="mask"):
        _lib.check(_lib.load().pxo_profile_enable(1 << 7), "pxo_profile_enable")
    with pytest.raises(_lib.PxoError, match="workspace"):
        ops.train_backward_work(pcfg, 8, torch.empty(1024, dtype=torch.uint8, device=dev))
    bad = pxo_cfg(ops, cfg); bad.skip_zero_rows = 2
    with pytest.raises(_lib.PxoError, match="skip_zero_rows"):
        ops.train_workspace_bytes(bad, 8)
    bad = pxo_cfg(ops, cfg); bad.noise_std = -0.5
    with pytest.raises(_lib.PxoError, match="noise_std"):
        ops.train_workspace_bytes(bad, 8)


def test_empty_inputs_are_accepted():
    """Zero rows / rays / points: every entry point returns without touching its (NULL) buffers; shapes survive."""
    ops = _ops(); dev = _gpu()
    cfg = O.Cfg(); pcfg = pxo_cfg(ops, cfg)
    flat = make_params(cfg).to(dev)
    packed = [ops.pack_weights(pcfg, split_mlp(flat, cfg, i), need_bwd=False) for i in range(2)]
    e3 = torch.zeros(0, 3, device=dev)
    rgb, sig = ops.eval_points(pcfg, packed[1][0], e3)
    assert rgb.shape == (0, 48) and sig.shape == (0, 1)
    out = ops.render_fwd(pcfg, packed[0][0], packed[1][0], e3, e3, e3)
    assert len(out) == 2 and out[1][0].shape == (0, 3) and out[1][1].shape == (0,)
    assert ops.posenc(e3).shape[0] == 0
    assert ops.uniform(1, 0, 0).numel() == 0 and ops.randint(1, 0, 0, 10).numel() == 0
    o, d, v = ops.generate_rays(torch.eye(4, device=dev)[:3], 8, 8, 10.0, pixel_ids=torch.zeros(0, dtype=torch.int64, device=dev))
    assert o.shape == (0, 3) and d.shape == (0, 3) and v.shape == (0, 3)
    assert ops.grid_sigma(pcfg, packed[1][0], 16, 5, 5, [0.0] * 3, [1.0] * 3).numel() == 0
    p = torch.zeros(0, device=dev)
    ops.adam_step(p, p.clone(), p.clone(), p.clone(), 1e-3, 1)
    assert ops.add_gaussian_noise(p.clone(), 0.5, seed=3).numel() == 0
    ob, db, vb, pxb = ops.sample_batch(1, 2, torch.eye(4, device=dev), 8, 8, 10.0, torch.zeros(64, 3, device=dev), 0)
    assert ob.shape == (0, 3) and pxb.shape == (0, 3)
    torch.cuda.synchronize()
    from plenoctree_amd import _lib
    with pytest.raises(_lib.PxoError):                      # the train step needs at least one ray
        ops.train_fwd_bwd(pcfg, flat, [ops.pack_weights(pcfg, split_mlp(flat, cfg, i)) for i in range(2)], e3, e3, e3, e3,
                          torch.zeros_like(flat), torch.zeros(6, device=dev), torch.empty(1 << 20, dtype=torch.uint8, device=dev))


def test_eval_points_past_2g_output_elements():
    """64-bit addressing: 46 M points x 48 coefficients = 2.2 G output floats (8.8 GB).  The evaluator is row-independent
    and bit-reproducible, so rows sampled from the big launch must equal the same points evaluated as a small batch."""
    ops = _ops(); dev = _gpu()
    cfg = O.Cfg(); pcfg = pxo_cfg(ops, cfg)
    flat = make_params(cfg)
    pf, _ = ops.pack_weights(pcfg, split_mlp(flat, cfg, 1).to(dev), need_bwd=False)
    N = 46_000_003
    pts = ops.uniform(11, 0, N * 3, -1.5, 1.5).reshape(N, 3)
    rgb, sig = ops.eval_points(pcfg, pf, pts)
    assert rgb.numel() > 2 ** 31
    pick = torch.cat([torch.arange(0, 300, device=dev), torch.arange(N - 300, N, device=dev),
                      torch.arange(44_739_243 - 150, 44_739_243 + 150, device=dev),        # around element 2^31
                      ops.randint(12, 0, 2000, N)])
    rgb_s, sig_s = ops.eval_points(pcfg, pf, pts[pick].contiguous())
    assert torch.equal(rgb[pick], rgb_s) and torch.equal(sig[pick], sig_s)
    assert bool(torch.isfinite(sig).all())


def test_bench_collective_path_through_rccl_single_rank():
    """The multi-GPU leg of bench.py on this 1-GPU box: launched the way the driver launches N > 1 (torch.distributed.run,
    one rank per GPU) with --force-dist, so the process group is RCCL and every step issues its two all-reduces (MLP_0's gradient on the
    side stream under the fine level, [MLP_1's gradient | stats] at the end) on the device; the throughput line must come
    out as without it."""
    _gpu()
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29517", os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2",
           "--batch", "1024", "--no-cpu-baseline", "--no-extras", "--force-dist"]
    res = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    line = [l for l in res.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 1 and out["nccl_ranks_seen"] == 1 and out["collectives_per_step"] == 2
    assert out["value"] > 5e4 and np.isfinite(out["final_stats"]["loss"])


def test_bucketed_gradient_exchange_rides_under_the_fine_level():
    """dist.GradReducer on a single-rank RCCL group (the only group a 1-GPU box can form): pxo_train_fwd_bwd_bucketed records
    `grads0_ready` after the coarse level's reverse, the side stream waits for that event only, and the all-reduce of
    MLP_0's half of the arena is therefore DONE before the step's last kernel is -- i.e. it ran under the fine level.
    The reduced step equals the unreduced one bit for bit (a sum over one rank), including with weight decay, whose MLP_0
    half must land before the event."""
    ops = _ops(); dev = _gpu()
    import torch.distributed as dist
    from plenoctree_amd import dist as pdist
    from plenoctree_amd.nerf_sh.nerf import models
    own = not dist.is_initialized()
    if own:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29519")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev, pg_options=pdist.nccl_options())
    try:
        cfg = O.Cfg(sparsity_npoints=1000, weight_decay_mult=0.1)
        pcfg = pxo_cfg(ops, cfg)
        B = 2048
        rays = make_rays(B)
        rays_dev = [r.to(dev) for r in rays]
        px = torch.rand(B, 3, generator=torch.Generator().manual_seed(1)).to(dev)
        results = []
        for use_reducer in (False, True):
            state = models.TrainState(pcfg, make_params(cfg).to(dev))
            model = models.NerfModel(pcfg)
            red = pdist.GradReducer(pdist.Comm(1, 0, 0, "nccl"), dev, force=True) if use_reducer else None
            if red is not None:
                assert red.active and red.side is not None
                red.record_timing = True
            batch = {"rays": type(rays)(*rays_dev), "pixels": px}
            for step in range(3):
                models.train_step(model, state, batch, 5e-4, randomized=True, seed=step, world_size=1, reducer=red)
            torch.cuda.synchronize()
            results.append((state.params.clone(), state.stats.clone()))
            if red is not None:
                # one more step by hand with an event at the end of the step's kernels
                ws = state.workspace(ops.train_workspace_bytes(pcfg, B))
                end = torch.cuda.Event(enable_timing=True)
                ops.train_fwd_bwd(pcfg, state.params, state.packed, *rays_dev, px, state.grads, state.stats, ws,
                                  randomized=True, seed=7, grads0_ready=red.ready_event())
                end.record()
                red.reduce(state.bucket0, state.bucket1)
                torch.cuda.synchronize()
                lead_ms = red.bucket0_done.elapsed_time(end)          # > 0: bucket 0 finished BEFORE the last kernel
                print(f"bucket 0 reduced {lead_ms:.3f} ms before the end of the step's kernels (B = {B})")
                assert lead_ms > 0.5, lead_ms                         # the fine level of 2048 rays takes ~9 ms
        assert torch.equal(results[0][0], results[1][0]) and torch.equal(results[0][1], results[1][1])
    finally:
        if own:
            dist.destroy_process_group()


@pytest.mark.parametrize("deg,shift,wd", [(3, -2.0, 0.0), (4, -1.0, 0.05), (3, -100.0, 0.0), (3, 0.0, 0.0)])
def test_skip_zero_rows_is_bit_identical(deg, shift, wd):
    """PxoCfg.skip_zero_rows: rows whose upstream gradient is exactly zero are left out of backward(data) (whole 128-row tiles)
    and of the weight-gradient GEMMs (16-row chunks).  Only exact zeros leave the sums, in unchanged order, so gradients
    and Stats must be BIT-identical to the dense pass -- with most of the volume empty (sigma-head biases lowered by `shift`
    x the spread of raw sigma), with everything empty (no live chunk at all) and with the ordinary mixture (shift 0)."""
    ops = _ops(); dev = _gpu()
    cfg = O.Cfg(sh_deg=deg, sparsity_npoints=777, weight_decay_mult=wd)
    flat = make_params(cfg)
    n = flat.numel() // 2
    b8 = sum(fi * fo + fo for fi, fo in O.layer_shapes(cfg)[:8]) + O.layer_shapes(cfg)[8][0]
    pts = (torch.rand(2048, 3, generator=torch.Generator().manual_seed(4)) * 2 - 1) * 2.0
    for mi in range(2):
        _, rs = O.mlp_forward(O.unflatten_params(flat, cfg)[mi], O.posenc(pts, 0, 10), cfg)
        flat[mi * n + b8] += shift * float(rs.std())
    B = 600                                        # coarse 38,400 rows = 300 tiles; fine 115,200 + 777 rows: ragged, half tiles
    rays = make_rays(B)
    px = torch.rand(B, 3, generator=torch.Generator().manual_seed(2))
    g = torch.Generator().manual_seed(3)
    t_rand, u = torch.rand(B, 64, generator=g), torch.rand(B, 128, generator=g)
    sp = (torch.rand(777, 3, generator=g) * 2 - 1) * 1.5
    outs = []
    for skip in (0, 1):
        pcfg = pxo_cfg(ops, cfg)
        pcfg.skip_zero_rows = skip
        fd = flat.to(dev)
        packed = [ops.pack_weights(pcfg, fd[i * n:(i + 1) * n].contiguous()) for i in range(2)]
        grads = torch.full_like(fd, float("nan")); stats = torch.zeros(6, device=dev)
        ws = torch.empty(ops.train_workspace_bytes(pcfg, B), dtype=torch.uint8, device=dev)
        ws.fill_(0xFF)                              # skipped tiles leave dz unwritten: NaN patterns there must never be read
        ops.train_fwd_bwd(pcfg, fd, packed, *[r.to(dev) for r in rays], px.to(dev), grads, stats, ws, randomized=True,
                          t_rand=t_rand.to(dev), u=u.to(dev), sp_points=sp.to(dev))
        live, total = ops.train_backward_work(pcfg, B, ws)
        outs.append((grads.cpu(), stats.cpu(), live, total))
    (g0, s0, l0, t0), (g1, s1, l1, t1) = outs
    print(f"deg {deg} shift {shift}: live chunks {l1}/{t1} = {l1 / t1:.3f}")
    assert bool(torch.isfinite(g0).all()) and bool(torch.isfinite(g1).all())
    assert torch.equal(g0, g1) and torch.equal(s0, s1)
    assert l0 == t0 == t1 == (B * 64 + 15) // 16 + (B * 192 + 777 + 15) // 16
    if shift <= -100.0:
        assert l1 == 0 and float(g1.abs().max()) == 0.0            # nothing is live: the reverse pass is skipped entirely
    elif shift < 0:
        assert 0 < l1 < 0.9 * t1, (l1, t1)                          # the case the option exists for (measured 0.32 / 0.84)
    else:
        assert l1 <= t1


def _one_train_step(ops, dev, cfg, flat, B, skip, seed=3):
    n = flat.numel() // 2
    rays = make_rays(B)
    px = torch.rand(B, 3, generator=torch.Generator().manual_seed(2))
    g = torch.Generator().manual_seed(seed)
    t_rand, u = torch.rand(B, 64, generator=g), torch.rand(B, 128, generator=g)
    sp = (torch.rand(cfg.sparsity_npoints, 3, generator=g) * 2 - 1) * 1.5
    pcfg = pxo_cfg(ops, cfg)
    pcfg.skip_zero_rows = skip
    fd = flat.to(dev)
    packed = [ops.pack_weights(pcfg, fd[i * n:(i + 1) * n].contiguous()) for i in range(2)]
    grads = torch.full_like(fd, float("nan")); stats = torch.zeros(6, device=dev)
    ws = torch.empty(ops.train_workspace_bytes(pcfg, B), dtype=torch.uint8, device=dev)
    ws.fill_(0xFF)
    ops.train_fwd_bwd(pcfg, fd, packed, *[r.to(dev) for r in rays], px.to(dev), grads, stats, ws, randomized=True,
                      t_rand=t_rand.to(dev), u=u.to(dev), sp_points=sp.to(dev))
    live, total = ops.train_backward_work(pcfg, B, ws)
    return grads.cpu(), stats.cpu(), live, total


@pytest.mark.parametrize("B,wd", [(512, 0.0), (2500, 0.05)])
def test_coarse_reverse_side_stream_is_bit_identical(B, wd):
    """PXO_TUNE_COARSE_REVERSE_STREAM: the coarse level's reverse pass on the library's side stream (beside the fine forward)
    or on the caller's stream -- the same kernels with the same inputs, so not a bit may change, step after step on ONE
    workspace (the slabs of the weight-gradient GEMMs are shared by the two levels: the join is what keeps them apart), dense and
    skipping, with weight decay (an axpy on the forked half) and with a NaN-poisoned workspace."""
    ops = _ops(); dev = _gpu()
    cfg = O.Cfg(sparsity_npoints=777, weight_decay_mult=wd)
    flat = make_params(cfg)
    default = ops.get_tuning(ops.TUNE_COARSE_REVERSE_STREAM)
    assert default == 0
    res = {}
    try:
        for mode in (0, 1):
            ops.set_tuning(ops.TUNE_COARSE_REVERSE_STREAM, mode)
            assert ops.get_tuning(ops.TUNE_COARSE_REVERSE_STREAM) == mode
            res[mode] = [_one_train_step(ops, dev, cfg, flat, B, skip, seed=3 + rep)[:2] for skip in (0, 1) for rep in range(2)]
    finally:
        ops.set_tuning(ops.TUNE_COARSE_REVERSE_STREAM, default)
    for (g0, s0), (g1, s1) in zip(res[0], res[1]):
        assert bool(torch.isfinite(g1).all())
        assert torch.equal(g0, g1) and torch.equal(s0, s1)


@pytest.mark.parametrize("B", [600, 2500])
def test_tile_counter_schedule_is_bit_identical(B):
    """PXO_TUNE_TILE_SCHED: the persistent workgroups of the dense training kernels take their tiles from a device counter
    instead of a static stride.  A slot's rows, relu-mask words and bias partial are a function of the slot alone, so gradients
    and Stats must not change by a bit -- B = 600: two rounds of tiles in the coarse pass, ragged half tiles in the fine one;
    B = 2500: 4 / 11 whole rounds + a half-tile round on 256 CUs -- and the skipping pass (always on the counter) must agree too."""
    ops = _ops(); dev = _gpu()
    cfg = O.Cfg(sparsity_npoints=777)
    flat = make_params(cfg)
    default = ops.get_tuning(ops.TUNE_TILE_SCHED)
    try:
        ops.set_tuning(ops.TUNE_TILE_SCHED, 0)
        g0, s0, _, _ = _one_train_step(ops, dev, cfg, flat, B, 0)
        ops.set_tuning(ops.TUNE_TILE_SCHED, 1)
        assert ops.get_tuning(ops.TUNE_TILE_SCHED) == 1
        g1, s1, _, _ = _one_train_step(ops, dev, cfg, flat, B, 0)
        g2, s2, _, _ = _one_train_step(ops, dev, cfg, flat, B, 1)
        g3, s3, _, _ = _one_train_step(ops, dev, cfg, flat, B, 0)       # again: the counters are re-zeroed by every step
    finally:
        ops.set_tuning(ops.TUNE_TILE_SCHED, default)
    assert bool(torch.isfinite(g0).all())
    for g, s_ in ((g1, s1), (g2, s2), (g3, s3)):
        assert torch.equal(g0, g) and torch.equal(s0, s_)


def test_wgrad_ranges_tuning_and_long_ranges_in_skipping_mode():
    """PXO_TUNE_WGRAD_RANGES / _SKINNY_RANGES change the split-K row ranges of the weight-gradient products: a different
    (still fixed) summation order, so results agree to float32 round-off, not bit for bit.  With row ranges longer than the
    skipping kernels' live-chunk list (32,768 rows: here forced with 2 ranges over 115,977 rows) the whole reverse pass must
    run DENSE -- the workspace is poisoned with NaNs, so a dense walk over the dz rows a skipping backward(data) left unwritten
    (what a silent per-kernel fallback would do) shows up as NaN gradients -- and must report every chunk as live."""
    ops = _ops(); dev = _gpu()
    cfg = O.Cfg(sparsity_npoints=777)
    flat = make_params(cfg)
    B = 600
    try:
        g0, s0, _, _ = _one_train_step(ops, dev, cfg, flat, B, 0)
        ops.set_tuning(ops.TUNE_WGRAD_RANGES, 73); ops.set_tuning(ops.TUNE_WGRAD_SKINNY_RANGES, 100)
        g1, s1, _, _ = _one_train_step(ops, dev, cfg, flat, B, 0)
        g1s, s1s, l1, t1 = _one_train_step(ops, dev, cfg, flat, B, 1)
        ops.set_tuning(ops.TUNE_WGRAD_RANGES, 2); ops.set_tuning(ops.TUNE_WGRAD_SKINNY_RANGES, 0)
        g2, s2, _, _ = _one_train_step(ops, dev, cfg, flat, B, 0)
        g2s, s2s, l2, t2 = _one_train_step(ops, dev, cfg, flat, B, 1)
    finally:
        ops.set_tuning(ops.TUNE_WGRAD_RANGES, 0); ops.set_tuning(ops.TUNE_WGRAD_SKINNY_RANGES, 0)
    assert torch.equal(s0, s1) and torch.equal(s0, s2)                     # the forward does not depend on the split
    for g in (g1, g2):
        assert bool(torch.isfinite(g).all())
        rel = float((g.double() - g0.double()).norm() / g0.double().norm())
        assert rel < 2e-6, rel
    assert torch.equal(g1, g1s) and l1 < t1                                 # skipping with ranges that fit: same bits, chunks skipped
    assert torch.equal(g2, g2s) and torch.equal(s2, s2s) and l2 == t2       # ranges too long for the live lists: dense, loudly so
    for knob, bad in ((ops.TUNE_TILE_SCHED, 2), (ops.TUNE_WGRAD_RANGES, -1), (ops.TUNE_WGRAD_RANGES, 100000), (99, 0)):
        with pytest.raises(Exception, match="pxo_set_tuning"):
            ops.set_tuning(knob, bad)


def test_per_host_image_shards_on_the_device():
    """datasets shard=(rank, world) through the fused launch (pxo_sample_batch's `first`): 8 x 512 rays = the 1 x 4096 batch,
    bit for bit, over several steps -- the reference's single-host sampler (datasets.py:159-166 + utils.shard)."""
    _ops(); dev = _gpu()
    from plenoctree_amd.nerf_sh.nerf import datasets, utils
    args = utils.define_flags().parse_args(["--config", "synthetic", "--train_dir", "x"])
    utils.update_flags(args); args.factor = 8
    whole = datasets.get_dataset("train", args, dev, batch_size=4096, seed=20201473)
    shards = [datasets.get_dataset("train", args, dev, batch_size=512, seed=20201473, shard=(r, 8)) for r in range(8)]
    for _ in range(3):
        want = next(whole)
        got = [next(d) for d in shards]
        assert torch.equal(torch.cat([g["pixels"] for g in got]), want["pixels"])
        for k in range(3):
            assert torch.equal(torch.cat([g["rays"][k] for g in got]), want["rays"][k])


def test_sample_batch_equals_the_three_separate_launches():
    """pxo_sample_batch (Dataset._next_train for one image in one launch) against pxo_randint + pxo_generate_rays + the gather
    of the image's colours: bit for bit, odd batch size included; and through datasets.Synthetic, whose batches must not
    change when the fused launch replaces the three."""
    ops = _ops(); dev = _gpu()
    from plenoctree_amd.nerf_sh.nerf.datasets import pose_spherical
    W, H, focal = 37, 29, 41.5
    c2w = torch.from_numpy(pose_spherical(33.0, 20.0, 4.0)).to(dev)
    image = torch.rand(H * W, 3, device=dev)
    for B in (1, 777, 4096):
        o, d, v, px, ids = ops.sample_batch(12345, 9, c2w, W, H, focal, image, B, want_ids=True)
        want_ids = ops.randint(12345, 9, B, W * H, device=dev)
        assert torch.equal(ids, want_ids)
        o2, d2, v2 = ops.generate_rays(c2w, W, H, focal, want_ids)
        assert torch.equal(o, o2) and torch.equal(d, d2) and torch.equal(v, v2)
        assert torch.equal(px, image[want_ids])
    from plenoctree_amd.nerf_sh.nerf import datasets, utils
    args = utils.define_flags().parse_args(["--config", "synthetic", "--train_dir", "x"])
    utils.update_flags(args); args.factor = 16
    fused = datasets.get_dataset("train", args, dev, batch_size=300)
    plain = datasets.get_dataset("train", args, dev, batch_size=300)
    class _NoFused:                       # the same feeder without the fused entry point
        def __init__(self, f): self.f = f
        def __getattr__(self, k):
            if k == "sample_batch":
                raise AttributeError(k)
            return getattr(self.f, k)
    plain.feeder = _NoFused(plain.feeder)
    for _ in range(3):
        a, b = next(fused), next(plain)
        assert torch.equal(a["pixels"], b["pixels"])
        for x, y in zip(a["rays"], b["rays"]):
            assert torch.equal(x, y)
