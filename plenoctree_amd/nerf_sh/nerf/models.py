"""NeRF-SH model and train step on the MI355X path.

Host-side mirror of the reference's nerf_sh/nerf/models.py (NerfModel.__call__ :216-348,
eval_points_raw :143-181, get_model_state :38-49) and nerf_sh/train.py:51-121 (train_step).
Parameters live in one flat float32 arena (MLP_0 then MLP_1, flax key order Dense_0..9);
all arithmetic is done by libplenoctree_hip.so through plenoctree_amd.ops.
"""
import math

import torch

from ... import ops
from . import utils


class TrainState:
    """optimizer.target + Adam moments + step (utils.TrainState, nerf_sh/nerf/utils.py:38-41),
    plus the MFMA-ordered weight images the kernels stream."""

    def __init__(self, cfg, params, step=0):
        self.cfg = cfg
        self.params = params
        self.m = torch.zeros_like(params)
        self.v = torch.zeros_like(params)
        self.step = int(step)            # number of updates applied (flax optimizer.state.step)
        # gradient arena with the 6 Stats in its tail: lax.pmean(grad) and lax.pmean(stats)
        # (train.py:117-118) are then ONE sum-all-reduce over `reduce_buf`
        self.reduce_buf = torch.zeros(params.numel() + 8, dtype=torch.float32, device=params.device)
        self.grads = self.reduce_buf[:params.numel()]
        self.stats = self.reduce_buf[params.numel():params.numel() + 6]
        self.n_mlp = params.numel() // 2
        # the two exchange buckets (dist.GradReducer): MLP_0's gradient | MLP_1's gradient + stats
        self.bucket0 = self.reduce_buf[:self.n_mlp]
        self.bucket1 = self.reduce_buf[self.n_mlp:]
        self.packed = [None, None]
        self._ws = None
        self.repack()

    def mlp_params(self, i):
        return self.params[i * self.n_mlp:(i + 1) * self.n_mlp]

    def repack(self, need_bwd=True):
        """Refresh the fragment-ordered images after a parameter update."""
        need_bwd = need_bwd and self.cfg.mlp_precision != ops._lib.MLP_BF16X3      # the bf16x3 images are forward-only
        for i in range(2):
            f, b = self.packed[i] if self.packed[i] is not None else (None, None)
            self.packed[i] = ops.pack_weights(self.cfg, self.mlp_params(i), f, b, need_bwd=need_bwd)

    def workspace(self, nbytes):
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = None
            self._ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=self.params.device)
        return self._ws


def glorot_uniform_(w, fan_in, fan_out, gen):
    lim = math.sqrt(6.0 / (fan_in + fan_out))
    w.copy_((torch.rand(w.shape, generator=gen, dtype=torch.float64) * 2 - 1).mul_(lim).float())


def init_params(cfg, seed=20200823):
    """Glorot-uniform kernels, zero biases (nerf_sh/nerf/model_utils.py:63-65), host RNG."""
    leaves, n = ops.param_layout(cfg)
    flat = torch.zeros(2 * n, dtype=torch.float32)
    gen = torch.Generator().manual_seed(seed)
    for mi in range(2):
        for layer, is_bias, off, rows, cols in leaves:
            if not is_bias:
                glorot_uniform_(flat[mi * n + off: mi * n + off + rows * cols].view(rows, cols), rows, cols, gen)
    return flat


class NerfModel:
    """Coarse + fine NeRF-SH renderer (nerf_sh/nerf/models.py:52-348)."""

    def __init__(self, cfg):
        self.cfg = cfg
        self.num_coarse_samples = cfg.num_coarse_samples
        self.num_fine_samples = cfg.num_fine_samples
        self.sh_deg = cfg.sh_deg

    def apply(self, state, rays, randomized, t_rand=None, u=None, seed=0):
        """model.apply(variables, key_0, key_1, rays, randomized) (:216): returns
        [(rgb, disp, acc)_coarse, (rgb, disp, acc)_fine].  The jax keys are replaced by
        explicit uniforms (t_rand [B,Nc], u [B,Nf]) or a Philox `seed`."""
        ws = state.workspace(ops.render_workspace_bytes(self.cfg, rays.origins.shape[0]))
        return ops.render_fwd(self.cfg, state.packed[0][0], state.packed[1][0], rays.origins, rays.directions,
                              rays.viewdirs, randomized=randomized, t_rand=t_rand, u=u, seed=seed, ws=ws)

    __call__ = apply

    def eval_points_raw(self, state, points, viewdirs=None, coarse=False, want_rgb=True):
        """:143-181 / octree/nerf/models.py:211-252: raw SH coefficients [N,3K] and raw sigma [N,1]."""
        which = 1 if (self.num_fine_samples > 0 and not coarse) else 0
        return ops.eval_points(self.cfg, state.packed[which][0], points, want_rgb=want_rgb)


def make_cfg(args):
    """PxoCfg from reference-named flags."""
    return ops.make_cfg(num_coarse_samples=args.num_coarse_samples, num_fine_samples=args.num_fine_samples,
                        sh_deg=args.sh_deg, min_deg_point=args.min_deg_point, max_deg_point=args.max_deg_point,
                        white_bkgd=int(args.white_bkgd), lindisp=int(args.lindisp),
                        sparsity_npoints=args.sparsity_npoints, near_=args.near, far_=args.far,
                        sparsity_weight=args.sparsity_weight, sparsity_length=args.sparsity_length,
                        sparsity_radius=args.sparsity_radius, weight_decay_mult=args.weight_decay_mult,
                        mlp_precision={"f32": 0, "bf16x3": 1, "bf16x6": 2}[getattr(args, "mlp_precision", "f32")],
                        noise_std=0.0 if getattr(args, "noise_std", None) is None else args.noise_std,
                        skip_zero_rows=int(bool(getattr(args, "skip_zero_rows", False))))


def construct_nerf(args, device, seed=None):
    """construct_nerf (:351-428): validates the configuration and initialises the parameters."""
    utils.check_supported(args)
    cfg = make_cfg(args)
    params = init_params(cfg, args.seed if seed is None else seed).to(device)
    return NerfModel(cfg), params


def get_model_state(args, device, restore=True):
    """get_model_state (:38-49): model + TrainState, restoring the newest checkpoint if any."""
    from . import checkpoints
    model, params = construct_nerf(args, device)
    state = TrainState(model.cfg, params)
    if restore and args.train_dir:
        checkpoints.restore_checkpoint(args.train_dir, state)
    return model, state


def train_step(model, state, batch, lr, randomized=True, t_rand=None, u=None, sp_points=None, seed=0,
               world_size=1, reducer=None):
    """One optimisation step (nerf_sh/train.py:51-121) on this rank's shard of the batch.

    loss_fn + value_and_grad run in pxo_train_fwd_bwd_bucketed; `reducer` (dist.GradReducer: two RCCL sums over the ranks,
    MLP_0's half of the gradient arena under the fine level, MLP_1's half with the 6 stats in its tail at the end)
    implements lax.pmean (train.py:117-118); Adam (train.py:119) and the re-pack of the weight images follow.  Returns
    the device tensor stats[6] = (loss, psnr, loss_c, loss_sp, psnr_c, weight_l2); its values are only read by the
    host when logging."""
    cfg = model.cfg
    rays = batch["rays"]
    B = rays.origins.shape[0]
    ws = state.workspace(ops.train_workspace_bytes(cfg, B))
    ev = reducer.ready_event() if reducer is not None else None
    ops.train_fwd_bwd(cfg, state.params, state.packed, rays.origins, rays.directions, rays.viewdirs, batch["pixels"],
                      state.grads, state.stats, ws, randomized=randomized, t_rand=t_rand, u=u, sp_points=sp_points,
                      seed=seed, grads0_ready=ev)
    scale = 1.0
    if reducer is not None:
        reducer.reduce(state.bucket0, state.bucket1)
    if world_size > 1:
        state.stats.mul_(1.0 / world_size)
        scale = 1.0 / world_size
    # Adam (train.py:119) and the refresh of the fragment-ordered weight images, one launch
    ops.adam_pack_step(cfg, state.params, state.m, state.v, state.grads, lr, state.step, state.packed, grad_scale=scale)
    state.step += 1
    return state.stats
