"""Oracle parity at the BASELINE.json sizes and constants (-m gpu).

configs[1]: SH16 blender preset, 4096 rays x (64 + 128) samples, 10k sparsity points;
configs[3]: SH25 tt preset (near 0, far 4, sparsity_length 0.2, sparsity_radius 5);
configs[4]: the 512^3 sigma grid of octree/extraction.py:290-320.
The oracle (oracle/nerf_oracle.py) is evaluated in float32 and, as the arbiter, in float64, over ray chunks
(tests/_helpers.py); on the GPU box's host cores this is a few seconds per evaluation.

Bounds (fixed numbers, not scaled by the CPU error):
  rendered colour     |dPSNR| <= 1e-4 dB against the f32 and the f64 oracle (north_star)
  Stats               rtol 2e-5
  gradient            relative L2 error vs the float64 oracle <= 2e-3 per MLP at 4096 rays, end to end (each path
                      draws its own fine samples; measured 0.9e-3 HIP / 0.7e-3 CPU-f32: the residual is the
                      ill-conditioned inverse-CDF step, which shrinks with the ray count -- 4e-2 at 48 rays, 1.5e-3 at
                      512; tests/test_gpu_parity.py::test_train_fwd_bwd_matches_oracle holds everything downstream of
                      the sample positions to 1e-3 at 48 rays)
"""
import os
import time

import numpy as np
import pytest
import torch

from oracle import nerf_oracle as O
from _helpers import (_gpu, _ops, _psnr, close, make_params, make_rays, oracle_loss_and_grad_chunked,
                      oracle_render_chunked, pxo_cfg, split_mlp)

pytestmark = pytest.mark.gpu

PRESETS = {
    # nerf_sh/config/blender.yaml over nerf_sh/nerf/utils.py:61-230
    "blender": dict(sh_deg=3, near=2.0, far=6.0, sparsity_length=0.05, sparsity_radius=1.5),
    # nerf_sh/config/tt.yaml
    "tt": dict(sh_deg=4, near=0.0, far=4.0, sparsity_length=0.2, sparsity_radius=5.0),
}
B_FULL = 4096
GRAD_BOUND = {4096: 2e-3, 512: 4e-3}
# Every HIP leg runs in native float32 AND in its opt-in float32-accurate emulation on the bf16 matrix pipe
# (PxoCfg.mlp_precision = bf16x6, csrc/mlp_x6_kernels.hip) -- at the SAME bounds.  The oracle legs are computed once per case.
PRECISIONS = [("f32", 0), ("bf16x6", 2)]
_ORACLE = {}


def _once(key, fn):
    if key not in _ORACLE:
        _ORACLE[key] = fn()
    return _ORACLE[key]


def _threads():
    torch.set_num_threads(min(32, os.cpu_count() or 1))


def _rays_for(preset, B, seed):
    rays = make_rays(B, seed)
    if preset == "tt":     # the tt scenes sit inside the near=0 .. far=4 range: cameras at distance 2
        rays = O.Rays(rays.origins * 0.5, rays.directions, rays.viewdirs)
    return rays


def _record(name, **kv):
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "fullsize_parity.jsonl"), "a") as f:
            f.write('{"test": "%s", %s}\n' % (name, ", ".join('"%s": %.6g' % (k, v) for k, v in kv.items())))


@pytest.mark.parametrize("prec_name,prec", PRECISIONS)
@pytest.mark.parametrize("preset,randomized", [("blender", True), ("blender", False), ("tt", True)])
def test_render_fwd_full_batch(preset, randomized, prec_name, prec):
    """pxo_render_fwd on 4096 rays against O.render (NerfModel.__call__, nerf_sh/nerf/models.py:216-348)."""
    ops = _ops(); dev = _gpu(); _threads()
    cfg = O.Cfg(**PRESETS[preset]); pcfg = pxo_cfg(ops, cfg)
    pcfg.mlp_precision = prec
    flat = make_params(cfg, bias_scale=0.2)
    B = B_FULL
    rays = _rays_for(preset, B, 141)
    gen = torch.Generator().manual_seed(143)
    t_rand = torch.rand(B, 64, generator=gen) if randomized else None
    u = torch.rand(B, 128, generator=gen) if randomized else None
    pk = [ops.pack_weights(pcfg, split_mlp(flat, cfg, i).to(dev), need_bwd=False)[0] for i in range(2)]
    out = ops.render_fwd(pcfg, pk[0], pk[1], rays.origins.to(dev), rays.directions.to(dev), rays.viewdirs.to(dev),
                         randomized=randomized, t_rand=None if t_rand is None else t_rand.to(dev),
                         u=None if u is None else u.to(dev))
    t0 = time.time()
    ref, ref64 = _once(("render", preset, randomized), lambda: (oracle_render_chunked(flat, rays, cfg, t_rand, u, torch.float32),
                                                                 oracle_render_chunked(flat, rays, cfg, t_rand, u, torch.float64)))
    t_cpu = time.time() - t0
    for lvl, tag in ((0, "coarse"), (1, "fine")):
        for j, name in ((0, "rgb"), (2, "acc")):
            got, r32, r64 = out[lvl][j].cpu().double(), ref[lvl][j].double(), ref64[lvl][j]
            assert torch.isfinite(got).all()
            # mean error against float64: HIP f32 must be of the same quality as the CPU f32 evaluation
            m_hip, m_cpu = float((got - r64).abs().mean()), float((r32 - r64).abs().mean())
            assert m_hip <= 3 * m_cpu + 2e-6, f"{tag}/{name}: mean err vs f64 {m_hip:.3g} (CPU f32: {m_cpu:.3g})"
    close("coarse/rgb (elementwise, well-conditioned stage)", out[0][0], ref[0][0], rtol=0, atol=3e-5)
    target = torch.rand(B, 3, generator=gen)
    p_hip, p32, p64 = _psnr(out[1][0].cpu(), target), _psnr(ref[1][0], target), _psnr(ref64[1][0], target)
    p_hip_c, p64_c = _psnr(out[0][0].cpu(), target), _psnr(ref64[0][0], target)
    _record(f"render_fwd[{preset},{randomized},{prec_name}]", psnr_hip=p_hip, psnr_f32=p32, psnr_f64=p64,
            d_f32=abs(p_hip - p32), d_f64=abs(p_hip - p64), oracle_s=t_cpu)
    assert abs(p_hip - p32) <= 1e-4 and abs(p_hip - p64) <= 1e-4, (p_hip, p32, p64)
    assert abs(p_hip_c - p64_c) <= 1e-4
    # and the image itself, arbitrated by the float64 render: the HIP image must be as close to it as the CPU float32
    # evaluation is (within 3 dB with deterministic sampling, 6 dB with random u: the residual sits in the handful of
    # pixels whose fine samples fall next to an empty stretch of the cdf, where one ulp of a coarse weight moves a sample
    # by a bin -- each float32 evaluation has its own few), with no pixel off by more than 1e-2 and at most 8 by > 1e-3.
    # Round 2 measured 76.3 dB here for deterministic sampling against 89.1 dB for the CPU: the float32 cdf of a ray that
    # ends on a surface plateaus at 1 - 2^-23, which IS the last deterministic u, and `u >= cdf` moved that sample to
    # the far end of the plateau; sample_pdf_kernel now accumulates the cdf in float64 (93.4 dB, profiles/r03_render_outliers.md).
    err = (out[1][0].cpu().double() - ref64[1][0]).abs().max(dim=-1)[0]
    p_img_hip, p_img_cpu = _psnr(out[1][0].cpu(), ref64[1][0]), _psnr(ref[1][0], ref64[1][0])
    _record(f"render_fwd_image[{preset},{randomized},{prec_name}]", psnr_hip_vs_f64=p_img_hip, psnr_f32_vs_f64=p_img_cpu,
            max_err=float(err.max()), n_over_1e4=float((err > 1e-4).sum()), n_over_1e3=float((err > 1e-3).sum()),
            median_err=float(err.median()))
    assert p_img_hip >= p_img_cpu - (6.0 if randomized else 3.0), (p_img_hip, p_img_cpu)
    assert p_img_hip >= 80.0
    assert float(err.max()) <= 1e-2 and int((err > 1e-3).sum()) <= 8, (float(err.max()), int((err > 1e-3).sum()))


@pytest.mark.parametrize("prec_name,prec", PRECISIONS)
@pytest.mark.parametrize("preset,wd", [("blender", 0.0), ("tt", 0.0), ("blender", 0.1)])
def test_train_fwd_bwd_full_batch(preset, wd, prec_name, prec):
    """pxo_train_fwd_bwd on one full step (4096 rays + 10k sparsity points) against loss_fn + value_and_grad
    (nerf_sh/train.py:68-116); wd > 0 exercises the weight_decay_mult term of the loss (train.py:101-114)."""
    ops = _ops(); dev = _gpu(); _threads()
    cfg = O.Cfg(weight_decay_mult=wd, **PRESETS[preset]); pcfg = pxo_cfg(ops, cfg)
    pcfg.mlp_precision = prec
    assert cfg.sparsity_npoints == 10000 and cfg.sparsity_weight == 1e-3
    flat = make_params(cfg, bias_scale=0.2)
    B = B_FULL if wd == 0.0 else 512
    rays = _rays_for(preset, B, 151)
    gen = torch.Generator().manual_seed(153)
    px = torch.rand(B, 3, generator=gen)
    t_rand = torch.rand(B, 64, generator=gen); u = torch.rand(B, 128, generator=gen)
    sp_pts = (torch.rand(cfg.sparsity_npoints, 3, generator=gen) * 2 - 1) * cfg.sparsity_radius
    fd = flat.to(dev)
    packed = [ops.pack_weights(pcfg, split_mlp(fd, cfg, i)) for i in range(2)]
    grads = torch.full_like(fd, float("nan")); stats = torch.zeros(6, device=dev)
    ws = torch.empty(ops.train_workspace_bytes(pcfg, B), dtype=torch.uint8, device=dev)
    ops.train_fwd_bwd(pcfg, fd, packed, rays.origins.to(dev), rays.directions.to(dev), rays.viewdirs.to(dev),
                      px.to(dev), grads, stats, ws, randomized=True, t_rand=t_rand.to(dev), u=u.to(dev),
                      sp_points=sp_pts.to(dev))
    t0 = time.time()
    (st32, g32), (st64, g64) = _once(("train", preset, wd), lambda: (
        oracle_loss_and_grad_chunked(flat, rays, px, cfg, t_rand, u, sp_pts, torch.float32),
        oracle_loss_and_grad_chunked(flat, rays, px, cfg, t_rand, u, sp_pts, torch.float64)))
    t_cpu = time.time() - t0
    s = stats.cpu()
    for i, k in enumerate(("loss", "psnr", "loss_c", "loss_sp", "psnr_c", "weight_l2")):
        close(f"stats/{k} (f32 oracle)", s[i], torch.tensor(st32[k]), rtol=2e-5, atol=1e-6)
        close(f"stats/{k} (f64 oracle)", s[i], torch.tensor(st64[k]), rtol=2e-5, atol=1e-6)
    n = flat.numel() // 2
    g_hip = grads.cpu().double()
    rec = {}
    for mi, (lo, hi) in enumerate(((0, n), (n, 2 * n))):
        ref = g64[lo:hi]
        assert float(ref.norm()) > 1e-4, "degenerate test: oracle gradient vanishes"
        e_hip = float((g_hip[lo:hi] - ref).norm() / ref.norm())
        e_cpu = float((g32[lo:hi].double() - ref).norm() / ref.norm())
        rec[f"mlp{mi}_hip"] = e_hip; rec[f"mlp{mi}_cpu"] = e_cpu
        assert e_hip <= GRAD_BOUND[B], f"MLP_{mi}: rel L2 err vs f64 oracle {e_hip:.3g} (CPU f32: {e_cpu:.3g})"
    _record(f"train_fwd_bwd[{preset},wd={wd},{prec_name}]", oracle_s=t_cpu, **rec)
    if wd > 0:      # the decay term alone: gradient difference between wd and 0 is 2*wd*p/n_params
        grads0 = torch.full_like(fd, float("nan"))
        cfg0 = O.Cfg(**PRESETS[preset])
        pcfg0 = pxo_cfg(ops, cfg0); pcfg0.mlp_precision = prec
        ops.train_fwd_bwd(pcfg0, fd, packed, rays.origins.to(dev), rays.directions.to(dev),
                          rays.viewdirs.to(dev), px.to(dev), grads0, stats, ws, randomized=True, t_rand=t_rand.to(dev),
                          u=u.to(dev), sp_points=sp_pts.to(dev))
        close("weight decay term", grads - grads0, 2 * wd * fd / fd.numel(), rtol=1e-3, atol=1e-9)


@pytest.mark.parametrize("prec_name,prec", PRECISIONS)
def test_grid_sigma_512_sample(prec_name, prec):
    """pxo_grid_sigma at reso 512 (134,217,728 voxels; octree/extraction.py:290-320) against O.eval_points_raw on
    2^20 randomly chosen voxels whose coordinates follow the reference's grid formula (:294-303)."""
    ops = _ops(); dev = _gpu(); _threads()
    cfg = O.Cfg(); pcfg = pxo_cfg(ops, cfg)
    pcfg.mlp_precision = prec
    flat = make_params(cfg, bias_scale=0.2)
    pf, _ = ops.pack_weights(pcfg, split_mlp(flat, cfg, 1).to(dev), need_bwd=False)
    reso = 512
    radius, center = torch.tensor([1.4, 1.5, 1.3]), torch.tensor([0.1, 0.0, -0.2])
    scale = 0.5 / radius; offset = 0.5 * (1.0 - center / radius)      # svox N3Tree(radius, center): invradius, offset
    sig = ops.grid_sigma(pcfg, pf, reso, 0, reso, offset.tolist(), scale.tolist())
    assert sig.shape == (reso ** 3,)
    assert bool(torch.isfinite(sig).all())
    gen = torch.Generator().manual_seed(7)
    idx = torch.randint(0, reso ** 3, (1 << 20,), generator=gen)
    idx[:4] = torch.tensor([0, reso ** 3 - 1, reso - 1, reso * reso])          # corners / stride boundaries
    arr = (torch.arange(0, reso, dtype=torch.float32) + 0.5) / reso             # extraction.py:294
    xx, yy, zz = [(arr - offset[i]) / scale[i] for i in range(3)]               # :295-297
    ix, iy, iz = idx // (reso * reso), (idx // reso) % reso, idx % reso         # meshgrid 'ij', x slowest (:303)
    pts = torch.stack([xx[ix], yy[iy], zz[iz]], -1)
    params = O.unflatten_params(flat, cfg)
    ref = torch.cat([O.eval_points_raw(params, pts[i:i + 65536], cfg)[1][:, 0] for i in range(0, pts.shape[0], 65536)])
    close("grid512 sigma", sig[idx.to(dev)], ref)
    # x-slab sharding at size (the 8-GPU split of config 5): slab 3 of 8 equals the same rows of the full grid
    slab = ops.grid_sigma(pcfg, pf, reso, 3 * 64, 4 * 64, offset.tolist(), scale.tolist())
    assert torch.equal(slab, sig[3 * 64 * reso * reso:4 * 64 * reso * reso])
