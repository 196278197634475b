"""CPU restatement of the reference NeRF-SH hot path (torch, float32 or float64).

TEST INFRASTRUCTURE -- the checker, never the product. Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

Parity status
-------------
* posenc / MLP / eval_sh are PINNED: tests/golden/*.npz were produced by
  importing the reference's own torch modules (octree/nerf/model_utils.py,
  octree/nerf/models.py, nerf_sh/nerf/sh.py) with tests/golden/make_golden.py,
  and tests/test_oracle_golden.py checks this file against them.  generate_rays
  and compute_psnr are PINNED the same way (octree/nerf/utils.py imported with
  stub modules for absl.flags / cv2, which carry no arithmetic).
* sample_along_rays / cast_rays / volumetric_rendering / piecewise_constant_pdf /
  sample_pdf (and the jax posenc) are PINNED against the reference's own function
  bodies: tests/golden/make_golden.py imports nerf_sh/nerf/model_utils.py with numpy
  (float32 defaults) standing in for jax.numpy, the random draws injected through
  the `key` argument and lax.stop_gradient = identity, and stores their outputs in
  tests/golden/model_utils.npz (executed by numpy instead of XLA; same source).
* the composition inside NerfModel.__call__ (`render`) is PINNED the same way:
  the reference's models.py + its MLP run through the shim with a dataclass stub of
  flax.linen.Module and a Dense stub fed the weights in flax's creation order
  (tests/golden/nerf_model.npz).
* add_gaussian_noise and its place in the composition (raw sigma, before the relu, separate draws for the two
  levels) are PINNED the same way with noise_std = 0.3 and injected normals (tests/golden/nerf_model_noise.npz,
  written by make_golden_grad.py's torch-backed shim).
* loss_fn's VALUE and Stats are PINNED against the reference's own train_step
  (nerf_sh/train.py:51-121) run through the shim with value_and_grad evaluating
  the function only (tests/golden/train_loss.npz).
* the GRADIENT of loss_fn is PINNED against reverse-mode AD through the reference's own
  loss_fn body: tests/golden/make_golden_grad.py imports nerf_sh/train.py, models.py,
  model_utils.py and sh.py from the reference with TORCH standing in for jax.numpy
  (jax.value_and_grad = torch autograd, lax.stop_gradient = detach) and stores the
  float64 gradient of a 24-ray step with sparsity and weight-decay terms in
  tests/golden/train_grad.npz; `loss_and_grad` reproduces it to the fixture's storage
  rounding, leaf by leaf (tests/test_oracle_golden.py).
* Adam (flax.optim) remains "PARITY UNPINNED": flax cannot be imported here and the
  reference ships no vectors for it; `adam_update` restates the published rule and is
  pinned only by closed-form known answers and by agreement with torch.optim.Adam, an independent
  implementation of the same published rule (tests/test_oracle_known_answers.py).
* flax.optim.Adam is third-party (flax>=0.3.1, environment.yml:19; call sites
  nerf_sh/nerf/models.py:44, nerf_sh/train.py:119); its published update rule
  is restated in `adam_update`.
* jax.random (threefry) streams are not reproducible; every random draw is an
  explicit argument (t_rand, u, sp_points).

All `file:line` citations are relative to /root/reference.
"""
import math
from collections import namedtuple

import numpy as np
import torch

Rays = namedtuple("Rays", ("origins", "directions", "viewdirs"))  # nerf_sh/nerf/utils.py:53


class Cfg:
    """Hyper-parameters of the path (defaults = nerf_sh/config/blender.yaml over
    nerf_sh/nerf/utils.py:61-230)."""

    def __init__(self, **kw):
        self.num_coarse_samples = 64
        self.num_fine_samples = 128
        self.sh_deg = 3
        self.near = 2.0
        self.far = 6.0
        self.net_depth = 8
        self.net_width = 256
        self.skip_layer = 4
        self.min_deg_point = 0
        self.max_deg_point = 10
        self.white_bkgd = True
        self.lindisp = False
        self.randomized = True
        self.sparsity_weight = 1e-3
        self.sparsity_length = 0.05
        self.sparsity_radius = 1.5
        self.sparsity_npoints = 10000
        self.weight_decay_mult = 0.0
        self.noise_std = None            # nerf_sh/nerf/utils.py:137-140 (no preset sets it)
        self.lr_init = 5e-4
        self.lr_final = 5e-6
        self.max_steps = 2000000
        self.lr_delay_steps = 0
        self.lr_delay_mult = 1.0
        for k, v in kw.items():
            if not hasattr(self, k):
                raise ValueError("unknown cfg key " + k)
            setattr(self, k, v)

    @property
    def sh_dim(self):
        return (self.sh_deg + 1) ** 2

    @property
    def num_rgb_channels(self):
        return 3 * self.sh_dim

    @property
    def input_dim(self):
        return 3 * (1 + 2 * (self.max_deg_point - self.min_deg_point))


# --------------------------------------------------------------------------
# parameters: a list (per MLP) of 10 (kernel[in,out], bias[out]) pairs, in the
# flax key order Dense_0..Dense_9 (octree/nerf/models.py:91-102): 0..7 trunk,
# 8 sigma head, 9 rgb head.
# --------------------------------------------------------------------------
def layer_shapes(cfg):
    d, w, inp = cfg.net_depth, cfg.net_width, cfg.input_dim
    shapes = []
    fan_in = inp
    for i in range(d):
        shapes.append((fan_in, w))
        # nerf_sh/nerf/model_utils.py:70-71: concat after layer i when i%skip==0, i>0
        fan_in = w + inp if (i % cfg.skip_layer == 0 and i > 0) else w
    shapes.append((fan_in, 1))                      # Dense_8 sigma (model_utils.py:72)
    shapes.append((fan_in, cfg.num_rgb_channels))   # Dense_9 rgb   (model_utils.py:91)
    return shapes


def init_mlp_params(cfg, gen, dtype=torch.float32):
    """Glorot-uniform kernels, zero bias (model_utils.py:63-65; torch twin
    octree/nerf/model_utils.py:28-33)."""
    params = []
    for fi, fo in layer_shapes(cfg):
        lim = math.sqrt(6.0 / (fi + fo))
        w = (torch.rand(fi, fo, generator=gen, dtype=torch.float64) * 2 - 1) * lim
        params.append((w.to(dtype), torch.zeros(fo, dtype=dtype)))
    return params


def init_params(cfg, seed=20200823, dtype=torch.float32):
    gen = torch.Generator().manual_seed(seed)
    return [init_mlp_params(cfg, gen, dtype), init_mlp_params(cfg, gen, dtype)]


def flatten_params(params):
    """Flat arena: MLP_0{Dense_0..9 kernel,bias}, MLP_1{...}; kernels row-major [in,out]."""
    out = []
    for mlp in params:
        for w, b in mlp:
            out.append(w.reshape(-1))
            out.append(b.reshape(-1))
    return torch.cat(out)


def unflatten_params(flat, cfg):
    params, off = [], 0
    for _ in range(2):
        mlp = []
        for fi, fo in layer_shapes(cfg):
            w = flat[off:off + fi * fo].reshape(fi, fo); off += fi * fo
            b = flat[off:off + fo]; off += fo
            mlp.append((w, b))
        params.append(mlp)
    assert off == flat.numel()
    return params


# --------------------------------------------------------------------------
# model_utils.py
# --------------------------------------------------------------------------
def cast_rays(z_vals, origins, directions):
    """nerf_sh/nerf/model_utils.py:97-101 (directions are NOT normalised)."""
    return origins[..., None, :] + z_vals[..., None] * directions[..., None, :]


def sample_along_rays(origins, directions, num_samples, near, far, t_rand, lindisp=False):
    """nerf_sh/nerf/model_utils.py:104-142. `t_rand` [B,S] in [0,1) replaces
    random.uniform(key, ...) (:135); None means randomized=False."""
    dt = origins.dtype
    B = origins.shape[0]
    t_vals = torch.linspace(0.0, 1.0, num_samples, dtype=dt)
    if lindisp:
        z_vals = 1.0 / (1.0 / near * (1.0 - t_vals) + 1.0 / far * t_vals)
    else:
        z_vals = near * (1.0 - t_vals) + far * t_vals
    if t_rand is not None:
        mids = 0.5 * (z_vals[..., 1:] + z_vals[..., :-1])
        upper = torch.cat([mids, z_vals[..., -1:]], -1)
        lower = torch.cat([z_vals[..., :1], mids], -1)
        z_vals = lower + (upper - lower) * t_rand
    else:
        z_vals = z_vals[None].expand(B, num_samples)
    return z_vals, cast_rays(z_vals, origins, directions)


def posenc(x, min_deg, max_deg):
    """nerf_sh/nerf/model_utils.py:145-173, default (non-legacy) order:
    [x | sin(xb) | sin(xb + pi/2)], xb index = l*3 + axis."""
    if min_deg == max_deg:
        return x
    scales = torch.tensor([2 ** i for i in range(min_deg, max_deg)], dtype=x.dtype)
    xb = (x[..., None, :] * scales[:, None]).reshape(list(x.shape[:-1]) + [-1])
    four_feat = torch.sin(torch.cat([xb, xb + 0.5 * math.pi], dim=-1))
    return torch.cat([x, four_feat], dim=-1)


def mlp_forward(mlp, x, cfg, return_acts=False):
    """nerf_sh/nerf/model_utils.py:43-94 with condition=None (use_viewdirs=false).
    x: [..., feature]; returns raw_rgb [...,3K], raw_sigma [...,1]."""
    lead = x.shape[:-1]
    x = x.reshape(-1, x.shape[-1])
    inputs = x
    acts = []
    for i in range(cfg.net_depth):
        w, b = mlp[i]
        x = torch.relu(x @ w + b)
        acts.append(x)
        if i % cfg.skip_layer == 0 and i > 0:
            x = torch.cat([x, inputs], dim=-1)
    ws, bs = mlp[cfg.net_depth]
    wr, br = mlp[cfg.net_depth + 1]
    raw_sigma = (x @ ws + bs).reshape(*lead, 1)
    raw_rgb = (x @ wr + br).reshape(*lead, wr.shape[1])
    if return_acts:
        return raw_rgb, raw_sigma, acts
    return raw_rgb, raw_sigma


# nerf_sh/nerf/sh.py:24-52
C0 = 0.28209479177387814
C1 = 0.4886025119029199
C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005,
      -1.0925484305920792, 0.5462742152960396]
C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
      -0.4570457994644658, 1.445305721320277, -0.5900435899266435]
C4 = [2.5033429417967046, -1.7701307697799304, 0.9461746957575601, -0.6690465435572892,
      0.10578554691520431, -0.6690465435572892, 0.47308734787878004, -1.7701307697799304,
      0.6258357354491761]


def sh_basis(deg, dirs):
    """The multipliers of sh[..., k] in nerf_sh/nerf/sh.py:72-108 as a [..., K] tensor."""
    x, y, z = dirs[..., 0], dirs[..., 1], dirs[..., 2]
    out = [torch.full_like(x, C0)]
    if deg > 0:
        out += [-C1 * y, C1 * z, -C1 * x]
    if deg > 1:
        xx, yy, zz = x * x, y * y, z * z
        xy, yz, xz = x * y, y * z, x * z
        out += [C2[0] * xy, C2[1] * yz, C2[2] * (2.0 * zz - xx - yy), C2[3] * xz,
                C2[4] * (xx - yy)]
    if deg > 2:
        out += [C3[0] * y * (3 * xx - yy), C3[1] * xy * z, C3[2] * y * (4 * zz - xx - yy),
                C3[3] * z * (2 * zz - 3 * xx - 3 * yy), C3[4] * x * (4 * zz - xx - yy),
                C3[5] * z * (xx - yy), C3[6] * x * (xx - 3 * yy)]
    if deg > 3:
        out += [C4[0] * xy * (xx - yy), C4[1] * yz * (3 * xx - yy), C4[2] * xy * (7 * zz - 1),
                C4[3] * yz * (7 * zz - 3), C4[4] * (zz * (35 * zz - 30) + 3),
                C4[5] * xz * (7 * zz - 3), C4[6] * (xx - yy) * (7 * zz - 1),
                C4[7] * xz * (xx - 3 * yy),
                C4[8] * (xx * (xx - 3 * yy) - yy * (3 * xx - yy))]
    return torch.stack(out, dim=-1)


def eval_sh(deg, sh, dirs):
    """nerf_sh/nerf/sh.py:54-109: sh [..., C, K], dirs [..., 3] -> [..., C]."""
    assert 0 <= deg <= 4 and (deg + 1) ** 2 == sh.shape[-1]
    return (sh * sh_basis(deg, dirs)[..., None, :]).sum(-1)


def volumetric_rendering(rgb, sigma, z_vals, dirs, white_bkgd):
    """nerf_sh/nerf/model_utils.py:176-222."""
    eps = 1e-10
    dists = torch.cat([z_vals[..., 1:] - z_vals[..., :-1],
                       torch.full_like(z_vals[..., :1], 1e10)], -1)
    dists = dists * torch.linalg.norm(dirs[..., None, :], dim=-1)
    alpha = 1.0 - torch.exp(-sigma[..., 0] * dists)
    accum_prod = torch.cat([torch.ones_like(alpha[..., :1]),
                            torch.cumprod(1.0 - alpha[..., :-1] + eps, dim=-1)], dim=-1)
    weights = alpha * accum_prod
    comp_rgb = (weights[..., None] * rgb).sum(dim=-2)
    depth = (weights * z_vals).sum(dim=-1)
    acc = weights.sum(dim=-1)
    inv_eps = 1 / eps
    disp = acc / depth
    disp = torch.where((disp > 0) & (disp < inv_eps) & (acc > eps), disp,
                       torch.full_like(disp, inv_eps))
    if white_bkgd:
        comp_rgb = comp_rgb + (1.0 - acc[..., None])
    return comp_rgb, disp, acc, weights


def piecewise_constant_pdf(bins, weights, num_samples, u):
    """nerf_sh/nerf/model_utils.py:225-286. `u` [B,num_samples] replaces
    random.uniform (:262); None means randomized=False (:264-266). No gradient."""
    dt = bins.dtype
    with torch.no_grad():
        eps = 1e-5
        weight_sum = weights.sum(dim=-1, keepdim=True)
        padding = torch.clamp(eps - weight_sum, min=0)
        weights = weights + padding / weights.shape[-1]
        weight_sum = weight_sum + padding
        pdf = weights / weight_sum
        cdf = torch.clamp(torch.cumsum(pdf[..., :-1], dim=-1), max=1)
        cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf, torch.ones_like(cdf[..., :1])], -1)
        if u is None:
            u = torch.linspace(0.0, 1.0 - float(np.finfo(np.float32).eps), num_samples, dtype=dt)
            u = u.expand(*cdf.shape[:-1], num_samples)
        mask = u[..., None, :] >= cdf[..., :, None]

        def find_interval(x):
            x0 = torch.where(mask, x[..., None], x[..., :1, None]).max(dim=-2)[0]
            x1 = torch.where(~mask, x[..., None], x[..., -1:, None]).min(dim=-2)[0]
            return x0, x1

        bins_g0, bins_g1 = find_interval(bins)
        cdf_g0, cdf_g1 = find_interval(cdf)
        t = torch.clamp(torch.nan_to_num((u - cdf_g0) / (cdf_g1 - cdf_g0), nan=0.0), 0, 1)
        return bins_g0 + t * (bins_g1 - bins_g0)


def sample_pdf(bins, weights, origins, directions, z_vals, num_samples, u):
    """nerf_sh/nerf/model_utils.py:289-314."""
    z_samples = piecewise_constant_pdf(bins, weights, num_samples, u)
    z_vals = torch.sort(torch.cat([z_vals, z_samples], dim=-1), dim=-1)[0]
    return z_vals, cast_rays(z_vals, origins, directions)


# --------------------------------------------------------------------------
# models.py
# --------------------------------------------------------------------------
def eval_points_raw(params, points, cfg, coarse=False):
    """nerf_sh/nerf/models.py:143-181 / octree/nerf/models.py:211-252 (no viewdirs)."""
    enc = posenc(points[None], cfg.min_deg_point, cfg.max_deg_point)
    mlp = params[1] if (cfg.num_fine_samples > 0 and not coarse) else params[0]
    raw_rgb, raw_sigma = mlp_forward(mlp, enc, cfg)
    return raw_rgb[0], raw_sigma[0]


def add_gaussian_noise(raw, noise_std, noise):
    """nerf_sh/nerf/model_utils.py:317-332: raw + normal * noise_std when noise_std is not None and randomized;
    `noise` (shaped like raw) replaces random.normal(key, raw.shape), None means randomized=False."""
    if noise_std is not None and noise is not None:
        return raw + noise.reshape(raw.shape) * noise_std
    return raw


def _shade(mlp, samples, viewdirs, cfg, noise=None):
    enc = posenc(samples, cfg.min_deg_point, cfg.max_deg_point)
    raw_rgb, raw_sigma = mlp_forward(mlp, enc, cfg)
    raw_sigma = add_gaussian_noise(raw_sigma, cfg.noise_std, noise)      # models.py:258-264 / :318-324
    # models.py:269-272: reshape(..., 3, K) then eval_sh with viewdirs[:, None]
    raw = eval_sh(cfg.sh_deg, raw_rgb.reshape(*raw_rgb.shape[:-1], -1, cfg.sh_dim),
                  viewdirs[:, None])
    rgb = torch.sigmoid(raw)          # models.py:280
    sigma = torch.relu(raw_sigma)     # models.py:281
    return rgb, sigma, raw_rgb, raw_sigma


def render(params, rays, cfg, t_rand=None, u=None, return_aux=False, noise_c=None, noise_f=None):
    """NerfModel.__call__, nerf_sh/nerf/models.py:216-348 (sh_deg>=0, no viewdirs).  noise_c [B,Nc] / noise_f [B,Nc+Nf]:
    the standard-normal draws of add_gaussian_noise when cfg.noise_std is set (None = randomized False).
    Returns [(rgb,disp,acc)_coarse, (rgb,disp,acc)_fine]."""
    aux = {}
    z_vals, samples = sample_along_rays(rays.origins, rays.directions, cfg.num_coarse_samples,
                                        cfg.near, cfg.far, t_rand, cfg.lindisp)
    rgb, sigma, raw_rgb, raw_sigma = _shade(params[0], samples, rays.viewdirs, cfg, noise_c)
    comp_rgb, disp, acc, weights = volumetric_rendering(rgb, sigma, z_vals, rays.directions,
                                                        cfg.white_bkgd)
    ret = [(comp_rgb, disp, acc)]
    aux.update(z_c=z_vals, w_c=weights, raw_rgb_c=raw_rgb, raw_sigma_c=raw_sigma)
    if cfg.num_fine_samples > 0:
        z_mid = 0.5 * (z_vals[..., 1:] + z_vals[..., :-1])           # models.py:296
        z_vals, samples = sample_pdf(z_mid, weights[..., 1:-1], rays.origins, rays.directions,
                                     z_vals, cfg.num_fine_samples, u)  # models.py:298-307
        rgb, sigma, raw_rgb, raw_sigma = _shade(params[1], samples, rays.viewdirs, cfg, noise_f)
        comp_rgb, disp, acc, weights = volumetric_rendering(rgb, sigma, z_vals, rays.directions,
                                                            cfg.white_bkgd)
        ret.append((comp_rgb, disp, acc))
        aux.update(z_f=z_vals, w_f=weights, raw_rgb_f=raw_rgb, raw_sigma_f=raw_sigma)
    if return_aux:
        return ret, aux
    return ret


# --------------------------------------------------------------------------
# train.py / utils.py
# --------------------------------------------------------------------------
def compute_psnr(mse):
    """nerf_sh/nerf/utils.py:384-393."""
    return -10.0 * torch.log(mse) / math.log(10.0)


def learning_rate_decay(step, lr_init, lr_final, max_steps, lr_delay_steps=0, lr_delay_mult=1):
    """nerf_sh/nerf/utils.py:483-515."""
    if lr_delay_steps > 0:
        delay_rate = lr_delay_mult + (1 - lr_delay_mult) * np.sin(
            0.5 * np.pi * np.clip(step / lr_delay_steps, 0, 1))
    else:
        delay_rate = 1.0
    t = np.clip(step / max_steps, 0, 1)
    return delay_rate * np.exp(np.log(lr_init) * (1 - t) + np.log(lr_final) * t)


def loss_fn(params, rays, pixels, cfg, t_rand, u, sp_points):
    """nerf_sh/train.py:68-114. Returns (total, stats dict)."""
    ret = render(params, rays, cfg, t_rand, u)
    if cfg.sparsity_weight > 0.0:
        _, sp_sigma = eval_points_raw(params, sp_points, cfg)
        sp_sigma = torch.relu(sp_sigma)
        loss_sp = cfg.sparsity_weight * (1.0 - torch.exp(-cfg.sparsity_length * sp_sigma).mean())
    else:
        loss_sp = torch.zeros((), dtype=pixels.dtype)
    rgb = ret[-1][0]
    loss = ((rgb - pixels[..., :3]) ** 2).mean()
    psnr = compute_psnr(loss)
    if len(ret) > 1:
        loss_c = ((ret[0][0] - pixels[..., :3]) ** 2).mean()
        psnr_c = compute_psnr(loss_c)
    else:
        loss_c = torch.zeros((), dtype=pixels.dtype)
        psnr_c = torch.zeros((), dtype=pixels.dtype)
    leaves = [t for mlp in params for pair in mlp for t in pair]
    weight_l2 = sum((z ** 2).sum() for z in leaves) / sum(z.numel() for z in leaves)
    stats = dict(loss=loss, psnr=psnr, loss_c=loss_c, loss_sp=loss_sp, psnr_c=psnr_c,
                 weight_l2=weight_l2)
    return loss + loss_c + loss_sp + cfg.weight_decay_mult * weight_l2, stats


def loss_and_grad(flat_params, rays, pixels, cfg, t_rand, u, sp_points):
    """jax.value_and_grad(loss_fn) (train.py:116) on the flat arena."""
    flat = flat_params.detach().clone().requires_grad_(True)
    total, stats = loss_fn(unflatten_params(flat, cfg), rays, pixels, cfg, t_rand, u, sp_points)
    total.backward()
    return total.detach(), {k: v.detach() for k, v in stats.items()}, flat.grad.detach()


def adam_update(param, m, v, grad, lr, step, beta1=0.9, beta2=0.999, eps=1e-8):
    """flax.optim.Adam.apply_param_gradient (flax>=0.3.1, third-party, restated):
    t = step+1 with `step` the number of updates already applied."""
    m = beta1 * m + (1.0 - beta1) * grad
    v = beta2 * v + (1.0 - beta2) * grad * grad
    t = step + 1
    m_hat = m / (1.0 - beta1 ** t)
    v_hat = v / (1.0 - beta2 ** t)
    param = param - lr * m_hat / (torch.sqrt(v_hat) + eps)
    return param, m, v


def train_step(flat_params, m, v, step, rays, pixels, cfg, t_rand, u, sp_points, lr):
    """nerf_sh/train.py:51-121 on one device (pmean over one replica = identity)."""
    _, stats, grad = loss_and_grad(flat_params, rays, pixels, cfg, t_rand, u, sp_points)
    p, m, v = adam_update(flat_params, m, v, grad, lr, step)
    return p, m, v, stats, grad


def generate_rays(w, h, focal, camtoworlds):
    """nerf_sh/nerf/utils.py:545-589 (pinhole branch), numpy float32."""
    x, y = np.meshgrid(np.arange(w, dtype=np.float32), np.arange(h, dtype=np.float32),
                       indexing="xy")
    camera_dirs = np.stack([(x - w * 0.5) / focal, -(y - h * 0.5) / focal, -np.ones_like(x)],
                           axis=-1)
    c2w = camtoworlds[:, None, None, :3, :3]
    directions = np.matmul(c2w, camera_dirs[None, ..., None])[..., 0]
    origins = np.broadcast_to(camtoworlds[:, None, None, :3, -1], directions.shape)
    viewdirs = directions / np.linalg.norm(directions, axis=-1, keepdims=True)
    return Rays(origins=origins, directions=directions, viewdirs=viewdirs)
