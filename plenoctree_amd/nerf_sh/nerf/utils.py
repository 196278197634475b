"""Host-side utilities of the NeRF-SH path: flags, containers, lr schedule, PSNR, ray
generation, chunked image rendering.  Mirrors the call surface of the reference's
nerf_sh/nerf/utils.py (cited per function) with argparse in place of absl (not installed)."""
import argparse
import collections
import math
import os

import numpy as np
import torch
import yaml

Rays = collections.namedtuple("Rays", ("origins", "directions", "viewdirs"))          # utils.py:53
Stats = collections.namedtuple("Stats", ("loss", "psnr", "loss_c", "loss_sp", "psnr_c", "weight_l2"))  # :43-50


def namedtuple_map(fn, tup):
    return type(tup)(*map(fn, tup))


def _tristate(v):
    if isinstance(v, bool):
        return v
    if v.lower() == "auto":
        return "auto"
    return _bool(v)


def _bool(v):
    if isinstance(v, bool):
        return v
    if v.lower() in ("1", "true", "yes", "y"):
        return True
    if v.lower() in ("0", "false", "no", "n"):
        return False
    raise argparse.ArgumentTypeError(f"bad bool {v}")


def define_flags(parser=None):
    """Same flag names and defaults as nerf_sh/nerf/utils.py:61-230 (hot-path subset + CLI)."""
    p = parser or argparse.ArgumentParser()
    a = p.add_argument
    a("--train_dir", type=str, default=None)
    a("--data_dir", type=str, default=None)
    a("--config", type=str, default=None)
    a("--dataset", type=str, default="blender", choices=["blender", "nsvf", "synthetic"])
    a("--image_batching", type=_bool, default=False)
    a("--white_bkgd", type=_bool, default=True)
    a("--batch_size", type=int, default=1024)
    a("--factor", type=int, default=4)
    a("--model", type=str, default="nerf")
    a("--near", type=float, default=2.0)
    a("--far", type=float, default=6.0)
    a("--net_depth", type=int, default=8)
    a("--net_width", type=int, default=256)
    a("--weight_decay_mult", type=float, default=0.0)
    # not a reference flag: opt-in split precision of the fused MLP kernels.  bf16x3: forward only (eval, gen_video, extraction),
    # |dPSNR| <= 1e-4 dB; bf16x6: float32-accurate (each operand split exactly in three bf16 parts, six partial products),
    # training included.  The default and every reported throughput are float32.
    a("--mlp_precision", type=str, default="f32", choices=["f32", "bf16x3", "bf16x6"])
    # not a reference flag: leave sample rows whose upstream gradient is exactly zero (empty space, occluded samples,
    # background rays) out of the reverse pass -- bit-identical gradients, faster steps once the scene has empty space
    a("--skip_zero_rows", type=_bool, default=True)
    # not a reference flag: with N ranks on one node, sample like the reference on ONE host with N local devices -- one image
    # per step, its batch_size pixels drawn once and sharded over the ranks (datasets.py:159-166 + utils.shard) -- instead of
    # like N hosts (each rank its own image).  N x (batch_size / N) then replays the 1 x batch_size batches exactly.
    # Default auto: ranks that share ONE host sample like the reference's local devices (true); ranks on different hosts like
    # its hosts (false).  dist.per_host_image() resolves it.
    a("--per_host_image", type=_tristate, default="auto")
    # not reference flags: shape of the `synthetic` dataset (the analytic stand-in scene) -- image size with the same horizontal
    # field of view, number of train / test poses, colours on the 8-bit grid of a PNG.  With them a run on `synthetic` and a run
    # on the same scene exported to the reference's on-disk formats (scripts/export_scene.py) see the same bits.
    a("--synthetic_hw", type=int, nargs=2, default=None)
    a("--synthetic_views", type=int, nargs=2, default=None)
    a("--synthetic_8bit", type=_bool, default=False)
    a("--skip_layer", type=int, default=4)
    a("--num_rgb_channels", type=int, default=3)
    a("--num_sigma_channels", type=int, default=1)
    a("--randomized", type=_bool, default=True)
    a("--min_deg_point", type=int, default=0)
    a("--max_deg_point", type=int, default=10)
    a("--deg_view", type=int, default=4)
    a("--num_coarse_samples", type=int, default=64)
    a("--num_fine_samples", type=int, default=128)
    a("--use_viewdirs", type=_bool, default=True)
    a("--sh_deg", type=int, default=-1)
    a("--sg_dim", type=int, default=-1)
    a("--noise_std", type=float, default=None)
    a("--lindisp", type=_bool, default=False)
    a("--net_activation", type=str, default="relu")
    a("--rgb_activation", type=str, default="sigmoid")
    a("--sigma_activation", type=str, default="relu")
    a("--legacy_posenc_order", type=_bool, default=False)
    a("--lr_init", type=float, default=5e-4)
    a("--lr_final", type=float, default=5e-6)
    a("--lr_delay_steps", type=int, default=0)
    a("--lr_delay_mult", type=float, default=1.0)
    a("--max_steps", type=int, default=1000000)
    a("--save_every", type=int, default=10000)
    a("--print_every", type=int, default=1000)
    a("--render_every", type=int, default=20000)
    a("--gc_every", type=int, default=5000)
    a("--sparsity_weight", type=float, default=1e-3)
    a("--sparsity_length", type=float, default=0.05)
    a("--sparsity_radius", type=float, default=1.5)
    a("--sparsity_npoints", type=int, default=10000)
    a("--eval_once", type=_bool, default=True)
    a("--save_output", type=_bool, default=True)
    a("--chunk", type=int, default=8192)
    a("--approx_eval_skip", type=int, default=1)     # evaluate only every x images (utils.py:225-229)
    a("--seed", type=int, default=20200823)
    # reference flags of parts that are not built here (LLFF scenes, the view-conditioned vanilla-NeRF head): accepted with
    # the reference's defaults so that its command lines and presets parse; check_supported rejects what would use them
    a("--spherify", type=_bool, default=False)             # utils.py:89
    a("--render_path", type=_bool, default=False)          # :90-94
    a("--llffhold", type=int, default=8)                   # :95-100
    a("--net_depth_condition", type=int, default=1)        # :108
    a("--net_width_condition", type=int, default=128)      # :109
    return p


def update_flags(args):
    """YAML preset merged over the parsed flags; unknown keys raise (utils.py:233-244)."""
    if args.config is None:
        return
    pth = args.config if args.config.endswith(".yaml") else args.config + ".yaml"
    if not os.path.exists(pth):  # allow `--config blender` to pick the in-tree preset
        alt = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "config", os.path.basename(pth))
        if os.path.exists(alt):
            pth = alt
    with open(os.path.expanduser(pth), "r") as fin:
        configs = yaml.safe_load(fin)
    invalid = sorted(set(configs) - set(vars(args)))
    if invalid:
        raise ValueError(f"Invalid args {invalid} in {pth}.")
    vars(args).update(configs)


def check_flags(args, require_data=True, require_batch_size_div=False, world_size=1):
    """utils.py:247-253, plus the restrictions of the MI355X path."""
    if args.train_dir is None:
        raise ValueError("train_dir must be set. None set now.")
    if require_data and args.data_dir is None and args.dataset != "synthetic":
        raise ValueError("data_dir must be set. None set now.")
    if require_batch_size_div and args.batch_size % world_size != 0:
        raise ValueError("Batch size must be divisible by the number of devices.")
    check_supported(args)


def check_supported(args):
    """The HIP path builds the SH configurations of config/blender.yaml and config/tt.yaml."""
    bad = []
    if args.use_viewdirs:
        bad.append("use_viewdirs=true (vanilla NeRF head)")
    if args.sh_deg < 0 or args.sh_deg > 4:
        bad.append(f"sh_deg={args.sh_deg} (need 0..4)")
    if args.sg_dim > 0:
        bad.append("sg_dim>0 (spherical gaussians)")
    if (args.net_depth, args.net_width, args.skip_layer) != (8, 256, 4):
        bad.append("net_depth/net_width/skip_layer != 8/256/4")
    if (args.min_deg_point, args.max_deg_point) != (0, 10):
        bad.append("min/max_deg_point != 0/10")
    if args.noise_std is not None and args.noise_std < 0:
        bad.append("noise_std < 0")
    if getattr(args, "render_path", False) or getattr(args, "spherify", False):
        bad.append("render_path / spherify (LLFF scenes)")
    if args.legacy_posenc_order:
        bad.append("legacy_posenc_order")
    if (args.net_activation.lower(), args.rgb_activation.lower(), args.sigma_activation.lower()) != ("relu", "sigmoid", "relu"):
        bad.append("activations other than relu/sigmoid/relu")
    if bad:
        raise NotImplementedError("not built on the MI355X path: " + "; ".join(bad))


def learning_rate_decay(step, lr_init, lr_final, max_steps, lr_delay_steps=0, lr_delay_mult=1):
    """utils.py:483-515: log-linear decay with an optional sine warm-up."""
    if lr_delay_steps > 0:
        delay_rate = lr_delay_mult + (1 - lr_delay_mult) * np.sin(0.5 * np.pi * np.clip(step / lr_delay_steps, 0, 1))
    else:
        delay_rate = 1.0
    t = np.clip(step / max_steps, 0, 1)
    return float(delay_rate * np.exp(np.log(lr_init) * (1 - t) + np.log(lr_final) * t))


def compute_psnr(mse):
    """utils.py:384-393."""
    return -10.0 * math.log(float(mse)) / math.log(10.0)


def compute_ssim(img0, img1, max_val=1.0, filter_size=11, filter_sigma=1.5, k1=0.01, k2=0.03):
    """Mean SSIM of [H,W,C] images (octree/nerf/utils.py:322-398, modelled after tf.image.ssim): separable Gaussian
    window, zero padding, variances clipped at 0.  An image metric outside the hot path: plain torch ops."""
    import torch.nn.functional as F
    c = img0.shape[-1]
    a = img0.reshape(-1, *img0.shape[-3:]).permute(0, 3, 1, 2)
    b = img1.reshape(-1, *img1.shape[-3:]).permute(0, 3, 1, 2)
    hw = filter_size // 2
    shift = (2 * hw - filter_size + 1) / 2
    f_i = ((torch.arange(filter_size, device=a.device, dtype=a.dtype) - hw + shift) / filter_sigma) ** 2
    filt = torch.exp(-0.5 * f_i)
    filt = filt / filt.sum()

    def blur(z):
        # separable window as shifted sums (zero padding) -- no convolution library involved
        H, W = z.shape[-2:]
        zp = F.pad(z, (hw, hw, 0, 0))
        z = sum(filt[k] * zp[..., :, k:k + W] for k in range(filter_size))
        zp = F.pad(z, (0, 0, hw, hw))
        return sum(filt[k] * zp[..., k:k + H, :] for k in range(filter_size))

    mu0, mu1 = blur(a), blur(b)
    mu00, mu11, mu01 = mu0 * mu0, mu1 * mu1, mu0 * mu1
    s00 = (blur(a * a) - mu00).clamp(min=0.0)
    s11 = (blur(b * b) - mu11).clamp(min=0.0)
    s01 = blur(a * b) - mu01
    s01 = torch.sign(s01) * torch.minimum(torch.sqrt(s00 * s11), s01.abs())
    c1, c2 = (k1 * max_val) ** 2, (k2 * max_val) ** 2
    ssim_map = ((2 * mu01 + c1) * (2 * s01 + c2)) / ((mu00 + mu11 + c1) * (s00 + s11 + c2))
    return ssim_map.reshape(ssim_map.shape[0], -1).mean(dim=-1)


def pose_spherical(theta, phi, radius, up_axis=0):
    """Camera-to-world of a view on a sphere around the origin (utils.py:656-685, "from NeRF"): theta/phi in
    degrees; up_axis 0 = default (z up), 1..5 = z down / y up / y down / x up / x down (volrend's number keys)."""
    th, ph = theta / 180.0 * np.pi, phi / 180.0 * np.pi
    c2w = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, radius], [0, 0, 0, 1]], np.float32)
    rot_phi = np.array([[1, 0, 0, 0], [0, np.cos(ph), -np.sin(ph), 0], [0, np.sin(ph), np.cos(ph), 0], [0, 0, 0, 1]], np.float32)
    rot_theta = np.array([[np.cos(th), 0, -np.sin(th), 0], [0, 1, 0, 0], [np.sin(th), 0, np.cos(th), 0], [0, 0, 0, 1]], np.float32)
    c2w = rot_theta @ (rot_phi @ c2w)
    c2w = np.array([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]], np.float32) @ c2w
    if up_axis != 0:
        vec_up, vec_1 = np.zeros(3, np.float32), np.zeros(3, np.float32)
        up_dim = 2 - up_axis // 2
        vec_up[up_dim] = -1 if up_axis % 2 else 1
        vec_1[1 if up_dim == 0 else 0] = 1
        trans = np.eye(4, dtype=np.float32)
        trans[:3, 0], trans[:3, 1], trans[:3, 2] = vec_1, np.cross(vec_up, vec_1), vec_up
        c2w = trans @ c2w
    return c2w


def save_img(img, pth):
    """utils.py:469-480: [H,W,C] in [0,1] (clipped) -> PNG."""
    from PIL import Image
    arr = np.asarray(img.detach().cpu() if hasattr(img, "detach") else img)
    Image.fromarray((np.clip(arr, 0.0, 1.0) * 255.0).astype(np.uint8)).save(pth, "PNG")


def shard(x, world_size, rank):
    """This rank's slice of a global batch (utils.py:518-522 reshapes [n_dev, B/n_dev, ...])."""
    per = x.shape[0] // world_size
    return x[rank * per:(rank + 1) * per]


def generate_rays(w, h, focal, camtoworlds):
    """Pinhole rays for every pixel, utils.py:545-589 (float32 numpy on the host)."""
    x, y = np.meshgrid(np.arange(w, dtype=np.float32), np.arange(h, dtype=np.float32), indexing="xy")
    camera_dirs = np.stack([(x - w * 0.5) / focal, -(y - h * 0.5) / focal, -np.ones_like(x)], axis=-1)
    c2w = camtoworlds[:, None, None, :3, :3]
    directions = np.matmul(c2w, camera_dirs[None, ..., None])[..., 0]
    origins = np.broadcast_to(camtoworlds[:, None, None, :3, -1], directions.shape)
    viewdirs = directions / np.linalg.norm(directions, axis=-1, keepdims=True)
    return Rays(origins=origins, directions=directions, viewdirs=viewdirs)


def render_image(render_fn, rays, normalize_disp=False, chunk=8192, world_size=1, rank=0, gather=None):
    """Render all pixels of an image in chunks (utils.py:331-381).

    render_fn(Rays[n,3]) -> [(rgb,disp,acc)_coarse, (rgb,disp,acc)_fine]; the last entry is
    used.  With world_size > 1 every chunk is padded to a multiple of world_size by edge
    replication (:357-364), each rank renders its slice, and `gather` (an all-gather over
    ranks, the lax.all_gather of utils.py:703-706) reassembles it before the padding is cut."""
    height, width = rays[0].shape[:2]
    num_rays = height * width
    rays = namedtuple_map(lambda r: r.reshape(num_rays, -1), rays)
    results = []
    for i in range(0, num_rays, chunk):
        chunk_rays = namedtuple_map(lambda r: r[i:i + chunk], rays)
        n = chunk_rays[0].shape[0]
        padding = (world_size - n % world_size) % world_size
        if padding:
            chunk_rays = namedtuple_map(lambda r: torch.cat([r, r[-1:].expand(padding, -1)], 0), chunk_rays)
        per = chunk_rays[0].shape[0] // world_size
        mine = namedtuple_map(lambda r: r[rank * per:(rank + 1) * per].contiguous(), chunk_rays)
        out = render_fn(mine)[-1]
        if world_size > 1:
            out = [gather(x) for x in out]
        if padding:
            out = [x[:-padding] for x in out]
        results.append(out)
    rgb, disp, acc = [torch.cat(r, dim=0) for r in zip(*results)]
    if normalize_disp:
        disp = (disp - disp.min()) / (disp.max() - disp.min())
    return rgb.reshape(height, width, -1), disp.reshape(height, width, -1), acc.reshape(height, width, -1)
