"""Generate golden vectors by IMPORTING the reference's own python modules.

Run in the authoring container only (needs /root/reference):
    python tests/golden/make_golden.py
Writes small .npz fixtures next to this file; they travel to the GPU box, the
reference does not.  Reference modules used (the only ones importable without
jax/flax/absl/svox):
  octree/nerf/model_utils.py  (MLP, posenc)
  octree/nerf/models.py       (NerfModel.eval_points_raw)
  nerf_sh/nerf/sh.py          (eval_sh)
  octree/nerf/sh_proj.py      (EvalSH: the reference's second, independent real-SH evaluation)
and, with stub modules standing in for `absl.flags` and `cv2` (flag registration / resize only, no arithmetic):
  octree/nerf/utils.py        (generate_rays, compute_psnr)
  octree/nerf/datasets.py     (Blender and NSVF loaders, run on the tiny on-disk scenes of golden_scenes.py)
and, with numpy (float32 defaults) standing in for `jax.numpy`, the random draws passed in through the `key`
argument, `lax.stop_gradient` = identity and an empty `flax.linen` stub -- i.e. the REFERENCE'S OWN FUNCTION
BODIES executed by numpy instead of XLA:
  nerf_sh/nerf/model_utils.py (cast_rays, sample_along_rays, posenc, volumetric_rendering,
                               piecewise_constant_pdf, sample_pdf)
"""
import os
import sys
import importlib.util

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def main():
    sys.path.insert(0, REF)
    from octree.nerf import model_utils as ref_mu          # noqa: E402
    from octree.nerf import models as ref_models           # noqa: E402
    ref_sh = _load("ref_sh", os.path.join(REF, "nerf_sh/nerf/sh.py"))

    torch.manual_seed(20200823)
    rng = np.random.default_rng(7)

    # ---- posenc -----------------------------------------------------------
    x = torch.tensor(rng.uniform(-4.0, 4.0, size=(37, 3)), dtype=torch.float32)
    x[0] = 0.0
    enc = ref_mu.posenc(x, 0, 10)
    np.savez(os.path.join(HERE, "posenc.npz"), x=x.numpy(), enc=enc.numpy())

    # ---- eval_points_raw for SH16 and SH25 --------------------------------
    for deg in (3, 4):
        K = (deg + 1) ** 2
        model = ref_models.NerfModel(
            num_coarse_samples=64, num_fine_samples=128, use_viewdirs=False, sh_deg=deg,
            sg_dim=-1, num_rgb_channels=3 * K, num_sigma_channels=1)
        # non-zero biases so that the bias path is exercised
        with torch.no_grad():
            for p in model.parameters():
                if p.dim() == 1:
                    p.uniform_(-0.1, 0.1)
        pts = torch.tensor(rng.uniform(-1.5, 1.5, size=(96, 3)), dtype=torch.float32)
        with torch.no_grad():
            rgb_f, sig_f = model.eval_points_raw(pts)
            rgb_c, sig_c = model.eval_points_raw(pts, coarse=True)
        out = dict(points=pts.numpy(), raw_rgb_fine=rgb_f.numpy(), raw_sigma_fine=sig_f.numpy(),
                   raw_rgb_coarse=rgb_c.numpy(), raw_sigma_coarse=sig_c.numpy())
        # parameters in flax key order / kernel [in,out] (octree/nerf/models.py:75-102)
        for mi, mlp in enumerate((model.MLP_0, model.MLP_1)):
            layers = list(mlp.input_layers) + [mlp.sigma_layer, mlp.rgb_layer]
            for li, layer in enumerate(layers):
                out[f"MLP_{mi}.Dense_{li}.kernel"] = layer.weight.detach().numpy().T.copy()
                out[f"MLP_{mi}.Dense_{li}.bias"] = layer.bias.detach().numpy().copy()
        np.savez_compressed(os.path.join(HERE, f"eval_points_sh{K}.npz"), **out)

    # ---- eval_sh deg 0..4 -------------------------------------------------
    out = {}
    d = rng.normal(size=(29, 3))
    d = d / np.linalg.norm(d, axis=-1, keepdims=True)
    out["dirs"] = d.astype(np.float32)
    for deg in range(5):
        K = (deg + 1) ** 2
        sh = rng.normal(size=(29, 5, 3, K)).astype(np.float32)
        res = ref_sh.eval_sh(deg, torch.tensor(sh), torch.tensor(out["dirs"])[:, None])
        out[f"sh_{deg}"] = sh
        out[f"res_{deg}"] = res.numpy()
    np.savez(os.path.join(HERE, "eval_sh.npz"), **out)

    # ---- generate_rays / compute_psnr / loaders (absl + cv2 stubbed) ---------
    import tempfile
    import types
    absl, flags, cv2 = types.ModuleType("absl"), types.ModuleType("absl.flags"), types.ModuleType("cv2")
    flags.FLAGS = types.SimpleNamespace()
    for name in ("DEFINE_string", "DEFINE_integer", "DEFINE_float", "DEFINE_bool", "DEFINE_enum", "DEFINE_boolean"):
        setattr(flags, name, lambda *a, **k: None)
    absl.flags = flags

    def cv2_resize(image, size, interpolation=None):
        """cv2.resize(..., INTER_AREA) for the exact integer down-scalings the fixtures use: on a divisible size the
        area filter IS the mean over factor x factor blocks (float32 accumulation order aside)."""
        ow, oh = size
        h, w = image.shape[:2]
        assert interpolation == "INTER_AREA" and h % oh == 0 and w % ow == 0 and h // oh == w // ow
        f = h // oh
        return image.reshape(oh, f, ow, f, -1).astype(np.float64).mean(axis=(1, 3)).astype(np.float32)
    cv2.resize, cv2.INTER_AREA = cv2_resize, "INTER_AREA"
    sys.modules.update({"absl": absl, "absl.flags": flags, "cv2": cv2})
    from octree.nerf import utils as ref_utils             # noqa: E402
    from octree.nerf import datasets as ref_datasets       # noqa: E402
    sys.path.insert(0, HERE)
    import golden_scenes                                   # noqa: E402

    c2w = golden_scenes.poses(3, seed=5)
    rays = ref_utils.generate_rays(9, 7, 12.5, c2w)
    mse = np.array([1e-3, 0.0371, 0.5], np.float32)
    rng_im = np.random.default_rng(99)                    # own stream: keeps the later fixtures unchanged
    im0 = torch.tensor(rng_im.uniform(size=(2, 24, 20, 3)), dtype=torch.float32)
    im1 = (im0 + 0.1 * torch.tensor(rng_im.normal(size=(2, 24, 20, 3)), dtype=torch.float32)).clamp(0, 1)
    np.savez(os.path.join(HERE, "generate_rays.npz"), c2w=c2w, w=9, h=7, focal=12.5, origins=rays.origins,
             directions=rays.directions, viewdirs=rays.viewdirs, mse=mse,
             psnr=np.array([float(ref_utils.compute_psnr(torch.tensor(m))) for m in mse]),
             ssim_im0=im0.numpy(), ssim_im1=im1.numpy(), ssim=ref_utils.compute_ssim(im0, im1, max_val=1.0).numpy())

    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        broot = golden_scenes.write_blender(os.path.join(tmp, "blender"))
        nroot = golden_scenes.write_nsvf(os.path.join(tmp, "nsvf"))
        for kind, root, cls in (("blender", broot, ref_datasets.Blender), ("nsvf", nroot, ref_datasets.NSVF)):
            for split in ("train", "test"):
                args = types.SimpleNamespace(data_dir=root, white_bkgd=True, factor=0, render_path=False,
                                             batch_size=4, image_batching=False)
                ds = cls(split, args)
                out[f"{kind}_{split}_images"] = np.asarray(ds.images, np.float32).reshape(ds.size, ds.h, ds.w, 3)
                out[f"{kind}_{split}_camtoworlds"] = np.asarray(ds.camtoworlds, np.float32)
                out[f"{kind}_{split}_hwf"] = np.array([ds.h, ds.w, ds.focal], np.float64)
        # factor = 2 (the only non-zero factor the reference's Blender loader accepts, datasets.py:100-109; NSVF: :430-444)
        for kind, root, cls in (("blender", broot, ref_datasets.Blender), ("nsvf", nroot, ref_datasets.NSVF)):
            args = types.SimpleNamespace(data_dir=root, white_bkgd=True, factor=2, render_path=False,
                                         batch_size=4, image_batching=False)
            ds = cls("train", args)
            out[f"{kind}_train_f2_images"] = np.asarray(ds.images, np.float32).reshape(ds.size, ds.h, ds.w, 3)
            out[f"{kind}_train_f2_hwf"] = np.array([ds.h, ds.w, ds.focal], np.float64)
    np.savez_compressed(os.path.join(HERE, "loaders.npz"), **out)

    # ---- nerf_sh/nerf/model_utils.py through a numpy-backed jax shim ----------
    jnp = types.ModuleType("jax.numpy")
    jnp.__dict__.update({k: v for k, v in np.__dict__.items() if not k.startswith("__")})
    f32 = np.float32
    jnp.linspace = lambda a, b, n: np.linspace(a, b, n, dtype=f32)          # jax default dtype is float32
    jnp.zeros = lambda shape, dtype=f32: np.zeros(shape, dtype)
    jnp.ones = lambda shape, dtype=f32: np.ones(shape, dtype)
    # jax promotes int32 * float32 -> float32 (numpy would give float64): keep small integer tables in float32 (exact)
    jnp.array = lambda x, dtype=None: np.array(x, dtype=dtype or f32)
    jax = types.ModuleType("jax")
    jrandom, lax, jnn = types.ModuleType("jax.random"), types.ModuleType("jax.lax"), types.ModuleType("jax.nn")
    jrandom.uniform = lambda key, shape: np.asarray(key, f32).reshape(shape)   # the draw IS the key
    jrandom.normal = lambda key, shape, dtype=f32: np.asarray(key, dtype).reshape(shape)
    lax.stop_gradient = lambda x: x
    jnn.initializers = types.SimpleNamespace(glorot_uniform=lambda: None)
    jax.numpy, jax.random, jax.lax, jax.nn = jnp, jrandom, lax, jnn
    flax, linen = types.ModuleType("flax"), types.ModuleType("flax.linen")
    linen.Module, linen.compact, linen.relu, linen.Dense = object, (lambda f: f), (lambda x: np.maximum(x, 0)), object
    flax.linen = linen
    sys.modules.update({"jax": jax, "jax.numpy": jnp, "jax.random": jrandom, "jax.lax": lax, "jax.nn": jnn,
                        "flax": flax, "flax.linen": linen})
    ref_jmu = _load("ref_jax_model_utils", os.path.join(REF, "nerf_sh/nerf/model_utils.py"))

    B, Nc, Nf = 7, 64, 128
    out = {}
    cam = rng.normal(size=(B, 3)); cam = (4.0 * cam / np.linalg.norm(cam, axis=-1, keepdims=True)).astype(f32)
    d = (-cam / 4.0 + 0.1 * rng.normal(size=(B, 3))).astype(f32)
    out["origins"], out["directions"] = cam, d
    out["t_rand"] = rng.uniform(size=(B, Nc)).astype(f32)
    for lindisp in (False, True):
        for randomized in (False, True):
            z, pts = ref_jmu.sample_along_rays(out["t_rand"], cam, d, Nc, 2.0, 6.0, randomized, lindisp)
            out[f"z_l{int(lindisp)}_r{int(randomized)}"] = np.asarray(z, f32)
            out[f"pts_l{int(lindisp)}_r{int(randomized)}"] = np.asarray(pts, f32)
    x = rng.uniform(-4.0, 4.0, size=(11, 3)).astype(f32)
    out["posenc_x"], out["posenc_enc"] = x, np.asarray(ref_jmu.posenc(x, 0, 10), f32)
    z = out["z_l0_r1"]
    rgb = rng.uniform(size=(B, Nc, 3)).astype(f32)
    sigma = np.maximum(rng.normal(size=(B, Nc, 1)) * 3.0, 0.0).astype(f32)
    sigma[0] = 0.0                                    # an empty ray
    sigma[1, 20:] = 1e4                               # an opaque surface
    out["vr_rgb"], out["vr_sigma"] = rgb, sigma
    for white in (False, True):
        comp, disp, acc, w = ref_jmu.volumetric_rendering(rgb, sigma, z, d, white)
        for name, val in (("comp", comp), ("disp", disp), ("acc", acc), ("weights", w)):
            out[f"vr_{name}_w{int(white)}"] = np.asarray(val, f32)
    w = out["vr_weights_w1"].copy()
    w[2] = 0.0                                        # all-zero weights: the eps-padding branch
    z_mid = 0.5 * (z[..., 1:] + z[..., :-1])          # nerf_sh/nerf/models.py:296-301
    out["pdf_bins"], out["pdf_weights"] = z_mid, w[..., 1:-1]
    out["pdf_u"] = rng.uniform(size=(B, Nf)).astype(f32)
    for randomized in (False, True):
        s = ref_jmu.piecewise_constant_pdf(out["pdf_u"], z_mid, w[..., 1:-1], Nf, randomized)
        out[f"pdf_samples_r{int(randomized)}"] = np.asarray(s, f32)
        zs, ps = ref_jmu.sample_pdf(out["pdf_u"], z_mid, w[..., 1:-1], cam, d, z, Nf, randomized)
        out[f"sample_pdf_z_r{int(randomized)}"] = np.asarray(zs, f32)
        out[f"sample_pdf_pts_r{int(randomized)}"] = np.asarray(ps, f32)
    np.savez_compressed(os.path.join(HERE, "model_utils.npz"), **out)

    # ---- NerfModel.__call__ (nerf_sh/nerf/models.py:216-348) through the same shim ---------
    # flax.linen stub: Modules are dataclasses whose setup() runs after init; Dense layers take their
    # (kernel, bias) from a queue in creation order = flax's auto-naming order Dense_0..Dense_9.
    import dataclasses
    from collections import namedtuple

    class Module:
        def __init_subclass__(cls, **kw):
            super().__init_subclass__(**kw)
            dataclasses.dataclass(cls, eq=False)

        def __post_init__(self):
            if hasattr(self, "setup"):
                self.setup()

    weight_queue = []

    class Dense:
        def __init__(self, features, kernel_init=None):
            self.features = features

        def __call__(self, x):
            kernel, bias = weight_queue.pop(0)
            assert kernel.shape == (x.shape[-1], self.features), (kernel.shape, x.shape, self.features)
            return x @ kernel + bias

    linen.Module, linen.Dense = Module, Dense
    linen.sigmoid = lambda x: (1.0 / (1.0 + np.exp(-x))).astype(f32)
    jrandom.split = lambda key, num=2: (key[0], key[1:])          # keys are lists of pre-drawn arrays
    ref_jmu = _load("nerf_sh.nerf.model_utils", os.path.join(REF, "nerf_sh/nerf/model_utils.py"))
    Rays = namedtuple("Rays", ("origins", "directions", "viewdirs"))
    pkg, pkg_nerf = types.ModuleType("nerf_sh"), types.ModuleType("nerf_sh.nerf")
    utils_stub, sg_stub = types.ModuleType("nerf_sh.nerf.utils"), types.ModuleType("nerf_sh.nerf.sg")
    utils_stub.Rays = Rays
    pkg.nerf = pkg_nerf
    pkg_nerf.model_utils, pkg_nerf.utils, pkg_nerf.sh, pkg_nerf.sg = ref_jmu, utils_stub, ref_sh, sg_stub
    sys.modules.update({"nerf_sh": pkg, "nerf_sh.nerf": pkg_nerf, "nerf_sh.nerf.utils": utils_stub,
                        "nerf_sh.nerf.sg": sg_stub, "nerf_sh.nerf.sh": ref_sh})
    ref_jmodels = _load("nerf_sh.nerf.models", os.path.join(REF, "nerf_sh/nerf/models.py"))
    gw = np.load(os.path.join(HERE, "eval_points_sh16.npz"))     # the weights already stored for the torch twin
    weights = [[(gw[f"MLP_{mi}.Dense_{li}.kernel"], gw[f"MLP_{mi}.Dense_{li}.bias"]) for li in range(10)]
               for mi in range(2)]
    # flax creation order inside MLP.__call__: Dense_0..7 trunk, then sigma (Dense_8), then rgb (Dense_9)
    model = ref_jmodels.NerfModel(
        num_coarse_samples=64, num_fine_samples=128, use_viewdirs=False, sh_deg=3, sg_dim=-1, near=2.0, far=6.0,
        noise_std=None, net_depth=8, net_width=256, net_depth_condition=1, net_width_condition=128,
        net_activation=linen.relu, skip_layer=4, num_rgb_channels=48, num_sigma_channels=1, white_bkgd=True,
        min_deg_point=0, max_deg_point=10, deg_view=4, lindisp=False, rgb_activation=linen.sigmoid,
        sigma_activation=linen.relu, legacy_posenc_order=False)
    B = 6
    cam = rng.normal(size=(B, 3)); cam = (4.0 * cam / np.linalg.norm(cam, axis=-1, keepdims=True)).astype(f32)
    d = (-cam / 4.0 + 0.08 * rng.normal(size=(B, 3))).astype(f32)
    v = (d / np.linalg.norm(d, axis=-1, keepdims=True)).astype(f32)
    t_rand, u = rng.uniform(size=(B, 64)).astype(f32), rng.uniform(size=(B, 128)).astype(f32)
    out = dict(origins=cam, directions=d, viewdirs=v, t_rand=t_rand, u=u)
    for randomized in (False, True):
        weight_queue[:] = weights[0] + weights[1]
        ret = model(*([t_rand, None], [u, None], Rays(cam, d, v), randomized))
        assert not weight_queue
        for lvl, (rgb_, disp_, acc_) in zip(("coarse", "fine"), ret):
            out[f"rgb_{lvl}_r{int(randomized)}"] = np.asarray(rgb_, f32)
            out[f"disp_{lvl}_r{int(randomized)}"] = np.asarray(disp_, f32)
            out[f"acc_{lvl}_r{int(randomized)}"] = np.asarray(acc_, f32)
    np.savez_compressed(os.path.join(HERE, "nerf_model.npz"), **out)

    # ---- loss_fn of train_step (nerf_sh/train.py:51-121): value + stats through the shim ----------
    # value_and_grad evaluates the function only (no AD here: gradients are checked by finite differences in the
    # oracle tests), pmean over one replica is the identity, the optimizer update is skipped.
    jrandom.split = lambda key, num=2: (key[0], key[1:]) if num == 2 else tuple(key[:num])
    jrandom.uniform = lambda key, shape, minval=0.0, maxval=1.0: (
        f32(minval) + (f32(maxval) - f32(minval)) * np.asarray(key, f32).reshape(shape)).astype(f32)
    jax.value_and_grad = lambda fn, has_aux=False: (lambda x: (fn(x), None))
    lax.pmean = lambda x, axis_name=None: x
    jax.lax = lax

    def tree_reduce(fn, tree, initializer=0):
        acc = initializer
        for mlp in tree:
            for kernel, bias in mlp:
                acc = fn(fn(acc, kernel), bias)
        return acc

    jax.tree_util = types.SimpleNamespace(tree_reduce=tree_reduce)
    jconfig = types.SimpleNamespace(parse_flags_with_absl=lambda: None)
    jax.config = jconfig
    absl.app = types.ModuleType("absl.app")
    Stats = namedtuple("Stats", ("loss", "psnr", "loss_c", "psnr_c", "weight_l2", "loss_sp"))
    utils_stub.Stats = Stats
    utils_stub.define_flags = lambda: None
    utils_stub.host0_print = print
    utils_stub.compute_psnr = lambda mse: -10.0 / np.log(10.0) * np.log(mse)      # nerf_sh/nerf/utils.py:384-393
    for name in ("flax.metrics", "flax.metrics.tensorboard", "flax.training", "flax.training.checkpoints",
                 "nerf_sh.nerf.datasets"):
        sys.modules[name] = types.ModuleType(name)
    flax.metrics = sys.modules["flax.metrics"]; flax.metrics.tensorboard = sys.modules["flax.metrics.tensorboard"]
    flax.training = sys.modules["flax.training"]; flax.training.checkpoints = sys.modules["flax.training.checkpoints"]
    pkg_nerf.datasets = sys.modules["nerf_sh.nerf.datasets"]
    pkg_nerf.models = ref_jmodels
    sys.modules.update({"absl.app": absl.app, "jax.config": jconfig})
    flags.FLAGS = types.SimpleNamespace(randomized=True, sparsity_weight=1e-3, sparsity_npoints=500, sparsity_radius=1.5,
                                        sparsity_length=0.05, weight_decay_mult=0.1)
    ref_train = _load("ref_nerf_sh_train", os.path.join(REF, "nerf_sh/train.py"))

    class ModelApply:                     # flax's model.apply(variables, *args, method=...)
        def apply(self, variables, *args, method=None):
            weight_queue[:] = [wb for mlp in variables for wb in mlp]
            if method is not None:        # eval_points_raw uses the fine MLP only: drop MLP_0's layers
                weight_queue[:] = list(variables[1])
                res = method(*args)
            else:
                res = model(*args)
            weight_queue[:] = []
            return res
        eval_points_raw = model.eval_points_raw

    state = types.SimpleNamespace(optimizer=types.SimpleNamespace(
        target=weights, apply_gradient=lambda grad, learning_rate: None), replace=lambda optimizer: None)
    pixels = rng.uniform(size=(B, 3)).astype(f32)
    sp_u = rng.uniform(size=(500, 3)).astype(f32)
    # rng -> (rng', key_0, key_1, key_2); key_2 is split once more before the sparsity draw (train.py:66,78-79)
    keys = [None, [t_rand, None], [u, None], [None, sp_u]]
    _, stats, _ = ref_train.train_step(ModelApply(), keys, state, {"rays": Rays(cam, d, v), "pixels": pixels}, 5e-4)
    out = dict(origins=cam, directions=d, viewdirs=v, t_rand=t_rand, u=u, pixels=pixels, sp_u=sp_u,
               sparsity_npoints=500, weight_decay_mult=0.1)
    out.update({k: np.float64(getattr(stats, k)) for k in Stats._fields})
    out["total"] = out["loss"] + out["loss_c"] + out["loss_sp"] + 0.1 * out["weight_l2"]
    np.savez(os.path.join(HERE, "train_loss.npz"), **out)

    # ---- nerf_sh/nerf/utils.py host helpers through the shim --------------------------------------
    flax.struct = types.SimpleNamespace(dataclass=dataclasses.dataclass)
    flax.optim = types.SimpleNamespace(Optimizer=object)
    for name in ("jax.dlpack", "jax.scipy"):
        sys.modules[name] = types.ModuleType(name)
    jax.dlpack, jax.scipy = sys.modules["jax.dlpack"], sys.modules["jax.scipy"]
    ref_jutils = _load("ref_nerf_sh_utils", os.path.join(REF, "nerf_sh/nerf/utils.py"))
    out = {}
    pose_args = [(30.0, -30.0, 4.0, 0), (-180.0, -30.0, 4.0, 0), (45.0, 10.0, 2.5, 1), (200.0, -60.0, 3.0, 2),
                 (10.0, -20.0, 4.0, 3), (10.0, -20.0, 4.0, 4), (10.0, -20.0, 4.0, 5)]
    out["pose_args"] = np.array(pose_args, np.float64)
    out["poses"] = np.stack([ref_jutils.pose_spherical(t, ph, r, int(up)) for t, ph, r, up in pose_args])
    lr_args = [(0, 5e-4, 5e-6, 2000000, 0, 1.0), (1000, 5e-4, 5e-6, 2000000, 0, 1.0), (2000000, 5e-4, 5e-6, 2000000, 0, 1.0),
               (3000000, 5e-4, 5e-6, 2000000, 0, 1.0), (10, 5e-4, 5e-6, 1000, 100, 0.01), (500, 1e-3, 1e-5, 1000, 100, 0.1)]
    out["lr_args"] = np.array(lr_args, np.float64)
    out["lr"] = np.array([float(ref_jutils.learning_rate_decay(int(a[0]), a[1], a[2], int(a[3]), int(a[4]), a[5]))
                          for a in lr_args])
    rays = ref_jutils.generate_rays(9, 7, 12.5, c2w)
    out["rays_origins"], out["rays_directions"], out["rays_viewdirs"] = [np.asarray(r, f32) for r in rays]
    np.savez(os.path.join(HERE, "nerf_sh_utils.npz"), **out)

    # ---- the reference's SECOND real-SH implementation: octree/nerf/sh_proj.py EvalSH(l, m, dirs) (:56-239) -----
    # (own generator: the draws above must not move)
    ref_proj = _load("ref_sh_proj", os.path.join(REF, "octree/nerf/sh_proj.py"))
    rng2 = np.random.default_rng(20200823)
    d = rng2.normal(size=(64, 3))
    d = (d / np.linalg.norm(d, axis=-1, keepdims=True)).astype(np.float64)
    basis = np.stack([np.asarray(ref_proj.EvalSH(l, m, torch.tensor(d))) for l in range(5) for m in range(-l, l + 1)], -1)
    np.savez(os.path.join(HERE, "sh_proj.npz"), dirs=d, basis=basis)
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
