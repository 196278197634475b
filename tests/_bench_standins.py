"""Oracle-backed CPU stand-ins for every HIP entry point bench.py reaches (TEST INFRASTRUCTURE: the product has no CPU
path).  tests/test_bench_dry_run_cpu.py installs them in each gloo rank before calling bench.main(), so that the file
the driver runs on 8 GPUs -- rendezvous, rank-0-only JSON line, the two gradient buckets per step, `strong512` inside an
existing process group, the all-gathers / max-all-reduce of the extraction records, cpu_baseline with world > 1 -- is
executed end to end on CPU ranks first.  Arithmetic comes from oracle/nerf_oracle.py and oracle/octree_oracle.py."""
import types

import numpy as np
import torch

from oracle import nerf_oracle as O
from oracle import octree_oracle as T


def _ocfg(pcfg):
    return O.Cfg(num_coarse_samples=pcfg.num_coarse_samples, num_fine_samples=pcfg.num_fine_samples, sh_deg=pcfg.sh_deg,
                 near=pcfg.near_, far=pcfg.far_, white_bkgd=bool(pcfg.white_bkgd), lindisp=bool(pcfg.lindisp),
                 sparsity_npoints=pcfg.sparsity_npoints, sparsity_weight=pcfg.sparsity_weight,
                 sparsity_length=pcfg.sparsity_length, sparsity_radius=pcfg.sparsity_radius,
                 weight_decay_mult=pcfg.weight_decay_mult)


def install():
    from plenoctree_amd import octree_ops as oops, ops
    from plenoctree_amd.nerf_sh.nerf import datasets
    from _cpu_feeder import feeder_for
    datasets.Dataset.feeder_factory = staticmethod(feeder_for)
    calls = {"train_fwd_bwd": 0, "render_fwd": 0, "grid_sigma": 0, "grid_weight_render": 0, "tree_from_mask": 0}

    def pack_weights(pcfg, mlp_params, f=None, b=None, need_bwd=True):
        return mlp_params, mlp_params                  # the stand-in images alias the parameters

    def _draws(cfg, B, seed, t_rand, u, sp):
        gen = torch.Generator().manual_seed(int(seed) & 0x7FFFFFFF)
        if t_rand is None:
            t_rand = torch.rand(B, cfg.num_coarse_samples, generator=gen)
        if u is None and cfg.num_fine_samples > 0:
            u = torch.rand(B, cfg.num_fine_samples, generator=gen)
        if sp is None:
            sp = (torch.rand(cfg.sparsity_npoints, 3, generator=gen) * 2 - 1) * cfg.sparsity_radius
        return t_rand, u, sp

    def train_fwd_bwd(pcfg, params, packed, o, d, v, px, grads, stats, ws, randomized=True, t_rand=None, u=None,
                      sp_points=None, seed=0, grads0_ready=None):
        assert grads0_ready is None
        calls["train_fwd_bwd"] += 1
        cfg = _ocfg(pcfg)
        t_rand, u, sp = _draws(cfg, o.shape[0], seed, t_rand, u, sp_points)
        _, st, g = O.loss_and_grad(params, O.Rays(o, d, v), px, cfg, t_rand, u, sp)
        grads.copy_(g)
        stats.copy_(torch.stack([st[k].float() for k in ("loss", "psnr", "loss_c", "loss_sp", "psnr_c", "weight_l2")]))

    def adam_pack_step(pcfg, params, m, v, grads, lr, step, packed, grad_scale=1.0):
        p, m2, v2 = O.adam_update(params, m, v, grads * grad_scale, lr, step)
        params.copy_(p); m.copy_(m2); v.copy_(v2)

    def render_fwd(pcfg, packed_fwd0, packed_fwd1, o, d, v, randomized=False, t_rand=None, u=None, seed=0, ws=None):
        calls["render_fwd"] += 1
        cfg = _ocfg(pcfg)
        if randomized:
            t_rand, u, _ = _draws(cfg, o.shape[0], seed, t_rand, u, 0)
        params = O.unflatten_params(torch.cat([packed_fwd0, packed_fwd1]), cfg)
        with torch.no_grad():
            return O.render(params, O.Rays(o, d, v), cfg, t_rand if randomized else None, u if randomized else None)

    def grid_sigma(pcfg, packed, reso, x0, x1, offset, scale, out=None):
        calls["grid_sigma"] += 1
        cfg = _ocfg(pcfg)
        arr = (torch.arange(reso, dtype=torch.float32) + 0.5) / reso
        xs = [(arr - float(offset[a])) / float(scale[a]) for a in range(3)]
        gx, gy, gz = torch.meshgrid(xs[0][x0:x1], xs[1], xs[2], indexing="ij")
        pts = torch.stack([gx, gy, gz], -1).reshape(-1, 3)
        mlp = O.unflatten_params(torch.cat([packed, packed]), cfg)[0]
        with torch.no_grad():
            _, rs = O.mlp_forward(mlp, O.posenc(pts, 0, 10), cfg)
        out.copy_(rs.reshape(-1))
        return out

    prof = {"on": False}
    ops.pack_weights, ops.train_fwd_bwd, ops.adam_pack_step = pack_weights, train_fwd_bwd, adam_pack_step
    ops.render_fwd, ops.grid_sigma = render_fwd, grid_sigma
    ops.train_workspace_bytes = lambda pcfg, B: 16
    ops.render_workspace_bytes = lambda pcfg, B: 16
    ops.profile_enable = lambda on=True, tags=None: prof.__setitem__("on", bool(on) if tags is None else bool(tags))
    ops.profile_read = lambda tag: (0, 0.0, 0)

    def render_opts(step_size=1e-3, background_brightness=1.0, sigma_thresh=0.0, stop_thresh=0.0):
        return types.SimpleNamespace(step_size=step_size, background_brightness=background_brightness,
                                     sigma_thresh=sigma_thresh, stop_thresh=stop_thresh)

    def grid_weight_render(sigma_grid, reso, c2w_all, fx, fy, width, height, opts, offset, invradius, grid_weight=None):
        # any camera-dependent, order-independent per-voxel quantity exercises the camera sharding + max-all-reduce
        calls["grid_weight_render"] += 1
        for c in c2w_all:
            grid_weight.copy_(torch.maximum(grid_weight, torch.sin(sigma_grid * float(c[:3, 3].sum())).abs()))
        return grid_weight

    def threshold_mask(value, thresh):
        return (value.reshape(-1) >= thresh).to(torch.uint8)

    def tree_from_mask(mask, depth):
        calls["tree_from_mask"] += 1
        reso = 2 ** (depth + 1)
        t = T.build_from_mask(mask.reshape(reso, reso, reso).numpy().astype(bool), depth, 1, [0.0, 0.0, 0.0], 1.0)
        levels = np.bincount(t.parent_depth[:, 1], minlength=depth + 1).tolist()
        return torch.from_numpy(t.child.copy()), torch.from_numpy(t.parent_depth.copy()), levels

    oops.render_opts, oops.grid_weight_render = render_opts, grid_weight_render
    oops.threshold_mask, oops.tree_from_mask = threshold_mask, tree_from_mask
    return calls
