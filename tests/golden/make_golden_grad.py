"""Gradient fixture from the REFERENCE'S OWN loss_fn: nerf_sh/train.py:51-121 (train_step), with
nerf_sh/nerf/models.py (NerfModel.__call__, eval_points_raw), nerf_sh/nerf/model_utils.py (MLP, posenc,
sample_along_rays, volumetric_rendering, piecewise_constant_pdf, sample_pdf) and nerf_sh/nerf/sh.py imported from
/root/reference and executed by TORCH standing in for `jax.numpy`:

  * jnp.*            -> thin numpy-signature wrappers over torch (float64 or float32 default dtype),
  * jax.random.*     -> the draw IS the key (pre-drawn arrays), split pops them in call order,
  * lax.stop_gradient-> Tensor.detach,
  * flax.linen       -> Module = dataclass stub, Dense = `x @ kernel + bias` on leaf tensors fed in flax's creation
                        order (Dense_0..7, sigma head Dense_8, rgb head Dense_9),
  * jax.value_and_grad(loss_fn, has_aux=True) -> torch autograd of the value loss_fn returns,
  * lax.pmean over one replica -> identity; optimizer.apply_gradient -> captures the gradient.

So the stored gradient is reverse-mode AD through the reference's function bodies, not through our restatement.
Run in the authoring container only (needs /root/reference):
    python tests/golden/make_golden_grad.py
Writes tests/golden/train_grad.npz and tests/golden/nerf_model_noise.npz (NerfModel.__call__ with noise_std = 0.3: F4).  Weights: those of eval_points_sh16.npz (reference torch twin, seed 20200823)
with the sigma-head bias of both MLPs raised by 0.5 so that every ray sees density (the loss then reaches every
parameter and the inverse-CDF sampling has no empty bins).
"""
import dataclasses
import importlib.util
import os
import sys
import types
from collections import namedtuple

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


class JT(torch.Tensor):
    """jax arrays are immutable: `weights += padding` (model_utils.py:243) REBINDS the name.  On a torch tensor the same
    statement would write through a view into the coarse weights that autograd still needs."""

    def __iadd__(self, other):
        return self + other


class Shim:
    """Installs the torch-backed jax/flax/absl stand-ins into sys.modules for one default dtype."""

    def __init__(self, dtype):
        self.dtype = dtype
        self.weight_queue = []
        self.captured = {}
        dt = dtype

        def T(x):
            if isinstance(x, torch.Tensor):
                return x
            return torch.as_tensor(np.asarray(x), dtype=dt).as_subclass(JT)
        self.T = T
        npdt = np.float64 if dt == torch.float64 else np.float32

        jnp = types.ModuleType("jax.numpy")
        jnp.ndarray, jnp.float32, jnp.pi = torch.Tensor, torch.float32, np.pi
        jnp.linspace = lambda a, b, n: T(np.linspace(a, b, n, dtype=npdt))
        jnp.array = lambda x, dtype=None: T(x)
        jnp.zeros = lambda shape, dtype=None: torch.zeros(tuple(shape), dtype=dt).as_subclass(JT)
        jnp.ones = lambda shape, dtype=None: torch.ones(tuple(shape), dtype=dt).as_subclass(JT)
        jnp.ones_like = lambda x, dtype=None: torch.ones_like(x)
        jnp.concatenate = lambda xs, axis=0: torch.cat([T(x) for x in xs], dim=axis)
        jnp.stack = lambda xs, axis=0: torch.stack([T(x) for x in xs], dim=axis)
        jnp.broadcast_to = lambda x, shape: T(x).expand(tuple(shape))
        jnp.reshape = lambda x, shape: x.reshape(tuple(shape))
        jnp.tile = lambda x, reps: x.repeat(*reps)
        jnp.sin, jnp.exp = torch.sin, torch.exp
        jnp.where = lambda c, a, b: torch.where(c, T(a), T(b))
        jnp.sum = lambda x, axis=None, keepdims=False: x.sum() if axis is None else x.sum(dim=axis, keepdim=keepdims)
        jnp.prod = lambda x: torch.prod(x)
        jnp.maximum = lambda a, b: torch.maximum(T(a), T(b))
        jnp.minimum = lambda a, b: torch.minimum(T(a), T(b))
        jnp.max = lambda x, axis: torch.max(x, dim=axis).values
        jnp.min = lambda x, axis: torch.min(x, dim=axis).values
        jnp.cumsum = lambda x, axis: torch.cumsum(x, dim=axis)
        jnp.cumprod = lambda x, axis: torch.cumprod(x, dim=axis)
        jnp.clip = lambda x, lo, hi: torch.clamp(x, lo, hi)
        # jnp.nan_to_num(x, 0): the second positional argument is `copy`; nan -> 0.0, +-inf -> largest finite
        jnp.nan_to_num = lambda x, copy=True: torch.nan_to_num(x)
        jnp.sort = lambda x, axis=-1: torch.sort(x, dim=axis).values
        jnp.any = lambda x: bool(torch.any(x))
        jnp.finfo = np.finfo
        jnp.linalg = types.SimpleNamespace(norm=lambda x, axis=None: torch.linalg.norm(x, dim=axis))
        jax = types.ModuleType("jax")
        jrandom, lax, jnn = types.ModuleType("jax.random"), types.ModuleType("jax.lax"), types.ModuleType("jax.nn")

        def split(key, num=2):                      # keys are lists of pre-drawn arrays (None where nothing is drawn)
            return (key[0], key[1:]) if num == 2 else tuple(key[:num])
        jrandom.split = split
        jrandom.uniform = lambda key, shape, minval=0.0, maxval=1.0: (minval + (maxval - minval) * T(key).reshape(tuple(shape)))
        jrandom.normal = lambda key, shape, dtype=None: T(key).reshape(tuple(shape))
        jrandom.PRNGKey = lambda seed: None
        lax.stop_gradient = lambda x: x.detach()
        lax.pmean = lambda x, axis_name=None: x
        jnn.initializers = types.SimpleNamespace(glorot_uniform=lambda: None)
        jnn.relu = torch.relu

        def value_and_grad(fn, has_aux=False):
            def run(target):
                leaves = [t for mlp in target for pair in mlp for t in pair]
                out = fn(target)
                total = out[0] if has_aux else out
                grads = torch.autograd.grad(total, leaves)
                it = iter(grads)
                tree = [[(next(it), next(it)) for _ in mlp] for mlp in target]
                return out, tree
            return run
        jax.value_and_grad = value_and_grad

        def tree_reduce(fn, tree, initializer=0):
            acc = initializer
            for mlp in tree:
                for kernel, bias in mlp:
                    acc = fn(fn(acc, kernel), bias)
            return acc
        jax.tree_util = types.SimpleNamespace(tree_reduce=tree_reduce)
        jax.config = types.SimpleNamespace(parse_flags_with_absl=lambda: None)
        jax.numpy, jax.random, jax.lax, jax.nn = jnp, jrandom, lax, jnn

        flax, linen = types.ModuleType("flax"), types.ModuleType("flax.linen")
        queue = self.weight_queue

        class Module:
            def __init_subclass__(cls, **kw):
                super().__init_subclass__(**kw)
                dataclasses.dataclass(cls, eq=False)

            def __post_init__(self):
                if hasattr(self, "setup"):
                    self.setup()

        class Dense:
            def __init__(self, features, kernel_init=None):
                self.features = features

            def __call__(self, x):
                kernel, bias = queue.pop(0)
                assert kernel.shape == (x.shape[-1], self.features), (kernel.shape, x.shape, self.features)
                return x @ kernel + bias

        linen.Module, linen.Dense, linen.compact = Module, Dense, (lambda f: f)
        linen.relu, linen.sigmoid = torch.relu, torch.sigmoid
        flax.linen = linen
        absl, flags, app = types.ModuleType("absl"), types.ModuleType("absl.flags"), types.ModuleType("absl.app")
        flags.FLAGS = types.SimpleNamespace()
        for name in ("DEFINE_string", "DEFINE_integer", "DEFINE_float", "DEFINE_bool", "DEFINE_enum", "DEFINE_boolean"):
            setattr(flags, name, lambda *a, **k: None)
        absl.flags, absl.app = flags, app
        self.flags = flags
        mods = {"jax": jax, "jax.numpy": jnp, "jax.random": jrandom, "jax.lax": lax, "jax.nn": jnn, "jax.config": jax.config,
                "flax": flax, "flax.linen": linen, "absl": absl, "absl.flags": flags, "absl.app": app}
        for name in ("flax.metrics", "flax.metrics.tensorboard", "flax.training", "flax.training.checkpoints",
                     "flax.jax_utils", "nerf_sh.nerf.datasets", "nerf_sh.nerf.sg"):
            mods[name] = types.ModuleType(name)
        flax.metrics, flax.training, flax.jax_utils = mods["flax.metrics"], mods["flax.training"], mods["flax.jax_utils"]
        flax.metrics.tensorboard = mods["flax.metrics.tensorboard"]
        flax.training.checkpoints = mods["flax.training.checkpoints"]
        sys.modules.update(mods)

        ref_sh = _load("nerf_sh.nerf.sh", os.path.join(REF, "nerf_sh/nerf/sh.py"))
        ref_mu = _load("nerf_sh.nerf.model_utils", os.path.join(REF, "nerf_sh/nerf/model_utils.py"))
        pkg, pkg_nerf = types.ModuleType("nerf_sh"), types.ModuleType("nerf_sh.nerf")
        utils_stub = types.ModuleType("nerf_sh.nerf.utils")
        self.Rays = namedtuple("Rays", ("origins", "directions", "viewdirs"))
        self.Stats = namedtuple("Stats", ("loss", "psnr", "loss_c", "psnr_c", "weight_l2", "loss_sp"))
        utils_stub.Rays, utils_stub.Stats = self.Rays, self.Stats
        utils_stub.define_flags = lambda: None
        utils_stub.host0_print = print
        utils_stub.compute_psnr = lambda mse: -10.0 / np.log(10.0) * torch.log(mse)      # nerf_sh/nerf/utils.py:384-393
        pkg.nerf = pkg_nerf
        pkg_nerf.model_utils, pkg_nerf.utils, pkg_nerf.sh = ref_mu, utils_stub, ref_sh
        pkg_nerf.sg, pkg_nerf.datasets = mods["nerf_sh.nerf.sg"], mods["nerf_sh.nerf.datasets"]
        sys.modules.update({"nerf_sh": pkg, "nerf_sh.nerf": pkg_nerf, "nerf_sh.nerf.utils": utils_stub})
        self.models = _load("nerf_sh.nerf.models", os.path.join(REF, "nerf_sh/nerf/models.py"))
        pkg_nerf.models = self.models
        self.train = _load("ref_nerf_sh_train_torch", os.path.join(REF, "nerf_sh/train.py"))
        self.linen = linen

    def model(self, sh_deg=3, noise_std=None):
        return self.models.NerfModel(
            num_coarse_samples=64, num_fine_samples=128, use_viewdirs=False, sh_deg=sh_deg, sg_dim=-1, near=2.0, far=6.0,
            noise_std=noise_std, net_depth=8, net_width=256, net_depth_condition=1, net_width_condition=128,
            net_activation=self.linen.relu, skip_layer=4, num_rgb_channels=3 * (sh_deg + 1) ** 2, num_sigma_channels=1,
            white_bkgd=True, min_deg_point=0, max_deg_point=10, deg_view=4, lindisp=False,
            rgb_activation=self.linen.sigmoid, sigma_activation=self.linen.relu, legacy_posenc_order=False)

    def train_step(self, weights_np, batch_np, t_rand, u, sp_u, fl):
        """The reference's train_step on one replica.  Returns (stats dict, gradient tree as numpy)."""
        T, queue = self.T, self.weight_queue
        self.flags.FLAGS.__dict__.update(fl)        # train.py binds FLAGS = flags.FLAGS at import
        model = self.model()
        leaves = [[(torch.tensor(k, dtype=self.dtype, requires_grad=True), torch.tensor(b, dtype=self.dtype, requires_grad=True))
                   for k, b in mlp] for mlp in weights_np]

        class ModelApply:                     # flax's model.apply(variables, *args, method=...)
            def apply(self_, variables, *args, method=None):
                if method is not None:        # eval_points_raw uses the fine MLP only
                    queue[:] = list(variables[1])
                    res = method(*args)
                else:
                    queue[:] = [wb for mlp in variables for wb in mlp]
                    res = model(*args)
                assert not queue
                return res
            eval_points_raw = model.eval_points_raw

        got = {}
        state = types.SimpleNamespace(optimizer=types.SimpleNamespace(
            target=leaves, apply_gradient=lambda grad, learning_rate: got.setdefault("grad", grad)),
            replace=lambda optimizer: None)
        rays = self.Rays(*[T(batch_np[k]) for k in ("origins", "directions", "viewdirs")])
        # rng -> (rng', key_0, key_1, key_2); key_2 is split once more before the sparsity draw (train.py:66,78-79)
        keys = [None, [t_rand, None], [u, None], [None, sp_u]]
        _, stats, _ = self.train.train_step(ModelApply(), keys, state, {"rays": rays, "pixels": T(batch_np["pixels"])}, 5e-4)
        grad = [[(k.detach().numpy(), b.detach().numpy()) for k, b in mlp] for mlp in got["grad"]]
        return {k: float(getattr(stats, k)) for k in self.Stats._fields}, grad


def noise_fixture(weights, rng):
    """NerfModel.__call__ with noise_std = 0.3 (models.py:258-264,318-324 -> model_utils.add_gaussian_noise :317-332):
    the two normal draws are injected through the keys, like the uniforms."""
    f32 = np.float32
    sh = Shim(torch.float32)
    model = sh.model(noise_std=0.3)
    B = 5
    cam = rng.normal(size=(B, 3)); cam = (4.0 * cam / np.linalg.norm(cam, axis=-1, keepdims=True)).astype(f32)
    d = (-cam / 4.0 + 0.08 * rng.normal(size=(B, 3))).astype(f32)
    v = (d / np.linalg.norm(d, axis=-1, keepdims=True)).astype(f32)
    t_rand, u = rng.uniform(size=(B, 64)).astype(f32), rng.uniform(size=(B, 128)).astype(f32)
    noise_c, noise_f = rng.normal(size=(B, 64)).astype(f32), rng.normal(size=(B, 192)).astype(f32)
    out = dict(origins=cam, directions=d, viewdirs=v, t_rand=t_rand, u=u, noise_c=noise_c, noise_f=noise_f, noise_std=0.3)
    leaves = [[(torch.tensor(k), torch.tensor(b)) for k, b in mlp] for mlp in weights]
    with torch.no_grad():
        for randomized in (False, True):
            sh.weight_queue[:] = [wb for mlp in leaves for wb in mlp]
            ret = model([t_rand, noise_c], [u, noise_f], sh.Rays(*[sh.T(x) for x in (cam, d, v)]), randomized)
            assert not sh.weight_queue
            for lvl, (rgb_, disp_, acc_) in zip(("coarse", "fine"), ret):
                out[f"rgb_{lvl}_r{int(randomized)}"] = rgb_.numpy().astype(f32)
                out[f"acc_{lvl}_r{int(randomized)}"] = acc_.numpy().astype(f32)
    np.savez_compressed(os.path.join(HERE, "nerf_model_noise.npz"), **out)
    print("wrote", os.path.join(HERE, "nerf_model_noise.npz"))


def flat_grad(grad):
    """flax key order Dense_0..9, kernel then bias, MLP_0 then MLP_1: the layout of pxo_param_layout / the oracle arena."""
    return np.concatenate([a.reshape(-1) for mlp in grad for pair in mlp for a in pair])


def main():
    sys.path.insert(0, REF)
    gw = np.load(os.path.join(HERE, "eval_points_sh16.npz"))
    weights = [[(gw[f"MLP_{mi}.Dense_{li}.kernel"].copy(), gw[f"MLP_{mi}.Dense_{li}.bias"].copy()) for li in range(10)]
               for mi in range(2)]
    for mi in range(2):
        weights[mi][8][1][:] += 0.5           # Dense_8 = sigma head
    rng = np.random.default_rng(20210302)
    f32 = np.float32
    B, n_sp = 24, 500
    cam = rng.normal(size=(B, 3)); cam = (4.0 * cam / np.linalg.norm(cam, axis=-1, keepdims=True)).astype(f32)
    d = (-cam / 4.0 + 0.08 * rng.normal(size=(B, 3))).astype(f32)
    v = (d / np.linalg.norm(d, axis=-1, keepdims=True)).astype(f32)
    batch = dict(origins=cam, directions=d, viewdirs=v, pixels=rng.uniform(size=(B, 3)).astype(f32))
    t_rand, u = rng.uniform(size=(B, 64)).astype(f32), rng.uniform(size=(B, 128)).astype(f32)
    sp_u = rng.uniform(size=(n_sp, 3)).astype(f32)
    fl = dict(randomized=True, sparsity_weight=1e-3, sparsity_npoints=n_sp, sparsity_radius=1.5, sparsity_length=0.05,
              weight_decay_mult=0.1)
    out = dict(batch, t_rand=t_rand, u=u, sp_u=sp_u, sigma_bias_shift=0.5, **{k: np.float64(val) for k, val in fl.items()})
    g = {}
    for tag, dt in (("f64", torch.float64), ("f32", torch.float32)):
        stats, grad = Shim(dt).train_step(weights, batch, t_rand, u, sp_u, fl)
        g[tag] = flat_grad(grad).astype(np.float64)
        for k, val in stats.items():
            out[f"{k}_{tag}"] = np.float64(val)
    rel = np.linalg.norm(g["f32"] - g["f64"]) / np.linalg.norm(g["f64"])
    print(f"gradient: {g['f64'].size} floats, |g| {np.linalg.norm(g['f64']):.6e}, reference-f32 vs reference-f64 rel L2 {rel:.3e}")
    out["grad"] = g["f64"].astype(f32)                     # float64 AD rounded once to float32 (4 MB)
    out["grad_norm_f64"] = np.float64(np.linalg.norm(g["f64"]))
    out["grad_f32_vs_f64_rel_l2"] = np.float64(rel)
    n = g["f64"].size // 2
    for mi in range(2):                                    # the same per MLP: the float32 noise floor of a 24-ray step
        a, b = g["f32"][mi * n:(mi + 1) * n], g["f64"][mi * n:(mi + 1) * n]
        out[f"grad_f32_vs_f64_rel_l2_mlp{mi}"] = np.float64(np.linalg.norm(a - b) / np.linalg.norm(b))
        print(f"  MLP_{mi}: reference-f32 vs reference-f64 rel L2 {float(out[f'grad_f32_vs_f64_rel_l2_mlp{mi}']):.3e}")
    np.savez_compressed(os.path.join(HERE, "train_grad.npz"), **out)
    print("wrote", os.path.join(HERE, "train_grad.npz"))
    plain = [[(gw[f"MLP_{mi}.Dense_{li}.kernel"].copy(), gw[f"MLP_{mi}.Dense_{li}.bias"].copy()) for li in range(10)]
             for mi in range(2)]
    noise_fixture(plain, np.random.default_rng(424242))


if __name__ == "__main__":
    main()
