"""Shared helpers of the -m gpu parity tests (HIP path through the C ABI vs the CPU oracle)."""
import os

import numpy as np   # noqa: F401
import pytest
import torch

from oracle import nerf_oracle as O


def _gpu():
    if not torch.cuda.is_available():
        pytest.fail("no ROCm GPU visible: -m gpu tests must run on the MI355X box")
    return torch.device("cuda:0")


def _ops():
    from plenoctree_amd import ops
    return ops


def close(name, got, want, rtol=2e-4, atol=2e-5):
    got = got.detach().cpu().double().reshape(-1)
    want = want.detach().cpu().double().reshape(-1)
    assert got.shape == want.shape, f"{name}: shape {got.shape} vs {want.shape}"
    assert torch.isfinite(got).all(), f"{name}: non-finite values in HIP output"
    assert torch.isfinite(want).all(), f"{name}: non-finite values in the oracle output"
    err = (got - want).abs()
    tol = atol + rtol * want.abs()
    bad = err > tol
    if bad.any():
        i = int(torch.argmax(err - tol))
        raise AssertionError(
            f"{name}: {int(bad.sum())}/{got.numel()} outside tol; worst idx {i}: got {got[i]:.8g} want {want[i]:.8g} "
            f"(abs err {err[i]:.3g}, max abs err {err.max():.3g}, ref max {want.abs().max():.3g})")


def make_params(cfg, seed=3, bias_scale=0.1, dtype=torch.float32):
    """Glorot kernels, N(0, bias_scale^2) biases, and a sigma head scaled so that rays see a mix
    of empty, translucent and opaque samples."""
    gen = torch.Generator().manual_seed(seed)
    params = [O.init_mlp_params(cfg, gen, dtype), O.init_mlp_params(cfg, gen, dtype)]
    out = []
    for mlp in params:
        for li, (w, b) in enumerate(mlp):
            if li == cfg.net_depth:          # Dense_8, sigma head
                w = w * 8.0
            out.append(w.reshape(-1))
            out.append(b + bias_scale * torch.randn(b.shape, generator=gen, dtype=dtype))
    flat = torch.cat(out)
    # A freshly initialised MLP has an almost constant raw sigma over space; shift each sigma-head
    # bias so that its median over the scene volume is slightly positive (otherwise relu(sigma) = 0
    # everywhere and every gradient vanishes).
    n = flat.numel() // 2
    b8 = sum(fi * fo + fo for fi, fo in O.layer_shapes(cfg)[:8]) + O.layer_shapes(cfg)[8][0]
    pts = (torch.rand(2048, 3, generator=gen, dtype=dtype) * 2 - 1) * 2.0
    for mi in range(2):
        mlp = O.unflatten_params(flat, cfg)[mi]
        _, rs = O.mlp_forward(mlp, O.posenc(pts, 0, 10), cfg)
        flat[mi * n + b8] += 0.2 * float(rs.std()) + 0.3 - float(rs.median())
    return flat


def make_rays(B, seed=5, dtype=torch.float32):
    gen = torch.Generator().manual_seed(seed)
    cam = torch.randn(B, 3, generator=gen, dtype=dtype)
    cam = 4.0 * cam / cam.norm(dim=-1, keepdim=True)
    target = 0.5 * (torch.rand(B, 3, generator=gen, dtype=dtype) - 0.5)
    d = target - cam
    d = d / d.norm(dim=-1, keepdim=True) * (1.0 + 0.1 * torch.rand(B, 1, generator=gen, dtype=dtype))
    v = d / d.norm(dim=-1, keepdim=True)
    return O.Rays(cam, d, v)


def pxo_cfg(ops, cfg):
    return ops.make_cfg(num_coarse_samples=cfg.num_coarse_samples, num_fine_samples=cfg.num_fine_samples,
                        sh_deg=cfg.sh_deg, white_bkgd=int(cfg.white_bkgd), lindisp=int(cfg.lindisp),
                        sparsity_npoints=cfg.sparsity_npoints, near_=cfg.near, far_=cfg.far,
                        sparsity_weight=cfg.sparsity_weight, sparsity_length=cfg.sparsity_length,
                        sparsity_radius=cfg.sparsity_radius, weight_decay_mult=cfg.weight_decay_mult)


def split_mlp(flat, cfg, which):
    n = flat.numel() // 2
    return flat[which * n:(which + 1) * n].contiguous()


def _psnr(a, b):
    return float(-10.0 * torch.log10(((a.double() - b.double()) ** 2).mean()))


def oracle_render_chunked(flat, rays, cfg, t_rand, u, dtype=torch.float32, chunk=1024):
    """O.render over ray chunks (bounds host memory at BASELINE sizes): [(rgb,disp,acc)_c, (rgb,disp,acc)_f]."""
    cast = lambda t: None if t is None else t.to(dtype)
    params = O.unflatten_params(flat.to(dtype), cfg)
    parts = []
    with torch.no_grad():
        for i0 in range(0, rays.origins.shape[0], chunk):
            sl = slice(i0, i0 + chunk)
            r = O.Rays(*[cast(x[sl]) for x in rays])
            parts.append(O.render(params, r, cfg, None if t_rand is None else cast(t_rand[sl]),
                                  None if u is None else cast(u[sl])))
    return [tuple(torch.cat([p[lvl][j] for p in parts]) for j in range(3)) for lvl in range(len(parts[0]))]


def oracle_loss_and_grad_chunked(flat, rays, px, cfg, t_rand, u, sp_points, dtype=torch.float32, chunk=512):
    """O.loss_fn + its gradient (nerf_sh/train.py:68-116) accumulated over ray chunks.  loss_fn's terms are means over
    rays, i.e. sums of per-chunk sums / (3B): the chunked evaluation is the same function (only the float summation
    order differs) with bounded host memory.  Returns (stats dict of floats, grad)."""
    cast = lambda t: None if t is None else t.to(dtype)
    p = flat.to(dtype).detach().clone().requires_grad_(True)
    B = px.shape[0]
    fine = cfg.num_fine_samples > 0
    sse = [0.0, 0.0]
    for i0 in range(0, B, chunk):
        sl = slice(i0, i0 + chunk)
        r = O.Rays(*[cast(x[sl]) for x in rays])
        ret = O.render(O.unflatten_params(p, cfg), r, cfg, cast(t_rand[sl]), cast(u[sl]) if fine else None)
        tgt = cast(px[sl])
        terms = [((lvl[0] - tgt) ** 2).sum() for lvl in ret]
        (sum(terms) / (3.0 * B)).backward()
        sse[0] += float(terms[-1].detach()); sse[1] += float(terms[0].detach()) if fine else 0.0
    params = O.unflatten_params(p, cfg)
    tail = torch.zeros((), dtype=dtype)
    loss_sp = 0.0
    if cfg.sparsity_weight > 0.0:
        _, sig = O.eval_points_raw(params, cast(sp_points), cfg)
        lsp = cfg.sparsity_weight * (1.0 - torch.exp(-cfg.sparsity_length * torch.relu(sig)).mean())
        tail = tail + lsp
        loss_sp = float(lsp.detach())
    leaves = [t for mlp in params for pair in mlp for t in pair]
    weight_l2 = sum((z ** 2).sum() for z in leaves) / sum(z.numel() for z in leaves)
    tail = tail + cfg.weight_decay_mult * weight_l2
    tail.backward()
    loss, loss_c = sse[0] / (3.0 * B), sse[1] / (3.0 * B)
    psnr = lambda m: -10.0 * float(np.log10(m)) if m > 0 else 0.0
    stats = dict(loss=loss, psnr=psnr(loss), loss_c=loss_c if fine else 0.0, loss_sp=loss_sp,
                 psnr_c=psnr(loss_c) if fine else 0.0, weight_l2=float(weight_l2))
    return stats, p.grad.detach()


def hip_sample_positions(ops, pcfg, packed, rays_dev, t_rand_dev, u_dev):
    """(z_coarse [B,Nc], z_fine [B,Nc+Nf]) exactly as the HIP path draws them: the same C-ABI kernels
    pxo_train_fwd_bwd sequences (sample_along_rays -> mlp_fwd -> shade_composite_fwd -> sample_pdf)."""
    o, d, v = rays_dev
    z_c, pts = ops.sample_along_rays(o, d, pcfg.num_coarse_samples, pcfg.near_, pcfg.far_, t_rand_dev)
    if pcfg.num_fine_samples == 0:
        return z_c, None
    raw_rgb, raw_sigma = ops.mlp_fwd(pcfg, packed[0][0], pts)
    _, _, _, w = ops.shade_composite_fwd(pcfg, raw_rgb, raw_sigma, z_c, d, v)
    z_f, _ = ops.sample_pdf(z_c, w, o, d, pcfg.num_fine_samples, u_dev)
    return z_c, z_f


def oracle_loss_and_grad_given_z(flat, rays, px, cfg, z_c, z_f, sp_points, dtype=torch.float64):
    """loss_fn (nerf_sh/train.py:68-114) and its gradient with the SAMPLE POSITIONS given instead of drawn.

    No gradient flows through the positions (lax.stop_gradient, nerf_sh/nerf/model_utils.py:286), so with z fixed
    this is the same differentiable function as O.loss_fn -- minus the ill-conditioned inverse-CDF step, whose own
    parity is tested through F(z) = u (test_sample_pdf) and through the rendered colours.  Composition follows
    O.render / nerf_sh/nerf/models.py:216-348 (cast_rays -> MLP -> eval_sh/sigmoid/relu -> volumetric_rendering)."""
    cast = lambda t: t.to(dtype)
    p = flat.to(dtype).detach().clone().requires_grad_(True)
    params = O.unflatten_params(p, cfg)
    o, d, v = [cast(x) for x in rays]
    tgt = cast(px)

    def level(mlp, z):
        z = cast(z)
        rgb, sigma, _, _ = O._shade(mlp, O.cast_rays(z, o, d), v, cfg)
        return O.volumetric_rendering(rgb, sigma, z, d, cfg.white_bkgd)[0]

    fine = cfg.num_fine_samples > 0
    rgb_c = level(params[0], z_c)
    loss_c = ((rgb_c - tgt) ** 2).mean()
    if fine:
        loss = ((level(params[1], z_f) - tgt) ** 2).mean()
    else:
        loss, loss_c = loss_c, torch.zeros((), dtype=dtype)
    total = loss + loss_c
    if cfg.sparsity_weight > 0.0:
        _, sig = O.eval_points_raw(params, cast(sp_points), cfg)
        total = total + cfg.sparsity_weight * (1.0 - torch.exp(-cfg.sparsity_length * torch.relu(sig)).mean())
    leaves = [t for mlp in params for pair in mlp for t in pair]
    total = total + cfg.weight_decay_mult * (sum((z ** 2).sum() for z in leaves) / sum(z.numel() for z in leaves))
    total.backward()
    return float(loss), float(loss_c), p.grad.detach()


def philox4x32_10(counter, key):
    """Philox4x32-10 (Salmon et al., SC'11; Random123) exactly as render_kernels.hip spells it: counter / key as tuples of
    32-bit words -> 4 output words.  Pinned to the Random123 known-answer vectors in tests/test_host_cpu.py."""
    m0, m1, w0, w1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
    c, (k0, k1) = list(counter), key
    for _ in range(10):
        p0, p1 = m0 * c[0], m1 * c[2]
        c = [((p1 >> 32) ^ c[1] ^ k0) & 0xFFFFFFFF, p1 & 0xFFFFFFFF, ((p0 >> 32) ^ c[3] ^ k1) & 0xFFFFFFFF, p0 & 0xFFFFFFFF]
        k0, k1 = (k0 + w0) & 0xFFFFFFFF, (k1 + w1) & 0xFFFFFFFF
    return c


def philox_uniform(seed, stream_id, n, lo=0.0, hi=1.0):
    """What pxo_uniform must return: element 4q+i = word i of block (counter = (q, stream_id), key = seed), top 24 bits."""
    import numpy as np
    out = np.empty(n, np.float32)
    for q in range((n + 3) // 4):
        w = philox4x32_10((q & 0xFFFFFFFF, q >> 32, stream_id & 0xFFFFFFFF, stream_id >> 32), (seed & 0xFFFFFFFF, seed >> 32))
        for i in range(4):
            if 4 * q + i < n:
                r01 = np.float32(w[i] >> 8) * np.float32(1.0 / 16777216.0)
                out[4 * q + i] = np.float32(lo) + (np.float32(hi) - np.float32(lo)) * r01
    return out


def philox_randint(seed, stream_id, count, n):
    """What pxo_randint must return: element 2q+i = (word 2i << 32 | word 2i+1) mod n of block q."""
    import numpy as np
    out = np.empty(count, np.int64)
    for q in range((count + 1) // 2):
        w = philox4x32_10((q & 0xFFFFFFFF, q >> 32, stream_id & 0xFFFFFFFF, stream_id >> 32), (seed & 0xFFFFFFFF, seed >> 32))
        for i in range(2):
            if 2 * q + i < count:
                out[2 * q + i] = ((w[2 * i] << 32) | w[2 * i + 1]) % n
    return out


# ---- the mid-scale trained-PSNR twin (tests/golden/make_trained_twin.py <-> test_trained_psnr_twin_512_rays) ------------
TWIN_RAYS, TWIN_STEPS = 512, 300
# the long twin (round 5): leaves the 14 dB regime -- HIP leg 23.9 dB on held-out views (scripts/twin_search.py,
# profiles/r05a_twin_search.jsonl: 1024 x 1500 -> 23.93, 1024 x 2000 -> 24.83, 2048 x 1000 -> 22.42 dB)
TWIN_LONG_RAYS, TWIN_LONG_STEPS = 1024, 1500


# the in-test twin of rounds 1-4 (test_trained_psnr_matches_oracle_training): 64 rays x 400 steps, 1000 sparsity points
TWIN_SHORT_RAYS, TWIN_SHORT_STEPS, TWIN_SHORT_SPARSITY = 64, 400, 1000


def twin_short_steps(cfg):
    """(step, host batch, t_rand, u, sp_points, lr) of the 64 x 400 twin: seeds only, so that the live oracle leg, the
    committed oracle leg (tests/golden/make_trained_twin.py short) and the HIP leg all replay the same inputs."""
    from plenoctree_amd.nerf_sh.nerf import datasets, utils
    B, steps = TWIN_SHORT_RAYS, TWIN_SHORT_STEPS
    ds = datasets.get_dataset("train", _twin_args(), torch.device("cpu"), batch_size=B)
    for step in range(steps):
        batch = next(ds)
        g = torch.Generator().manual_seed(1000 + step)
        t_rand = torch.rand(B, 64, generator=g); u = torch.rand(B, 128, generator=g)
        sp = (torch.rand(TWIN_SHORT_SPARSITY, 3, generator=g) * 2 - 1) * 1.5
        yield step, batch, t_rand, u, sp, utils.learning_rate_decay(step, 5e-4, 5e-6, steps)


def twin_short_digests(cfg):
    """What the committed oracle leg of the 64 x 400 twin (tests/golden/trained_twin_64x400.json) depends on, as two sha256
    digests: the oracle's SOURCE (oracle/nerf_oracle.py) and the INPUTS it was run on (initial parameters, every 50th step's
    batch / randoms / learning rate, the held-out rays and pixels) as this code generates them today.  The test compares them
    with the fixture's: a change to the oracle, the seeds, the sampler or the scene makes the fixture stale, and the test then
    trains the oracle live instead of comparing HIP with an oracle that no longer exists."""
    import hashlib
    here = os.path.dirname(os.path.abspath(__file__))
    with open(os.path.join(here, "..", "oracle", "nerf_oracle.py"), "rb") as f:
        code = hashlib.sha256(f.read()).hexdigest()
    h = hashlib.sha256()
    # values are hashed on a 2^-12 grid: the scene's pixels are float32 torch-CPU arithmetic, whose last bit may differ between
    # hosts (vector math libraries); a changed seed, sampler or scene moves them by far more
    q = lambda t: torch.round(t.double() * 4096.0).to(torch.int64).contiguous().numpy().tobytes()
    h.update(q(O.flatten_params(O.init_params(cfg, seed=20200823))))
    for step, batch, t_rand, u, sp, lr in twin_short_steps(cfg):
        if step % 50 == 0 or step == TWIN_SHORT_STEPS - 1:
            for t in (*batch["rays"], batch["pixels"], t_rand, u, sp):
                h.update(q(t))
            h.update(("%.9e" % float(lr)).encode())
    rays, px = twin_heldout()
    for t in (*rays, px):
        h.update(q(t))
    return {"oracle_sha256": code, "inputs_sha256": h.hexdigest()}


def _twin_args():
    from plenoctree_amd.nerf_sh.nerf import utils
    args = utils.define_flags().parse_args(["--config", "synthetic", "--train_dir", "x"])
    utils.update_flags(args); args.factor = 8             # 100 x 100 views of the analytic scene
    return args


def twin_steps(B, steps, cfg):
    """(step, host batch, t_rand, u, sp_points, lr) of the twin run: batches from the CPU feeder (np.random.RandomState, as
    the reference's sampler), randoms from per-step torch generators, the reference's log-linear lr schedule
    5e-4 -> 5e-6 annealed over the horizon (nerf_sh/nerf/utils.py:483-515).  Seeds only: both legs regenerate it."""
    from plenoctree_amd.nerf_sh.nerf import datasets, utils
    ds = datasets.get_dataset("train", _twin_args(), torch.device("cpu"), batch_size=B)
    for step in range(steps):
        batch = next(ds)
        g = torch.Generator().manual_seed(5000 + step)
        t_rand = torch.rand(B, cfg.num_coarse_samples, generator=g); u = torch.rand(B, cfg.num_fine_samples, generator=g)
        sp = (torch.rand(cfg.sparsity_npoints, 3, generator=g) * 2 - 1) * cfg.sparsity_radius
        yield step, batch, t_rand, u, sp, utils.learning_rate_decay(step, 5e-4, 5e-6, steps)


def twin_heldout():
    """Held-out rays / pixels: every 4th pixel of three views of the TEST split."""
    from plenoctree_amd.nerf_sh.nerf import datasets
    test_ds = datasets.get_dataset("test", _twin_args(), torch.device("cpu"))
    views = [test_ds.get_image(i) for i in (0, 67, 133)]
    rays = O.Rays(*[torch.cat([t["rays"][k].reshape(-1, 3)[::4] for t in views]).contiguous() for k in range(3)])
    px = torch.cat([t["pixels"].reshape(-1, 3)[::4] for t in views])
    return rays, px
