#!/usr/bin/env python
"""How far apart do two float32 evaluations of the long twin (1024 rays x 1500 steps) end, as a function of how much their
GRADIENTS differ per step?

Two float32 implementations of this training run never see the same gradient: the fine level's gradient has a condition number
of ~1e4 with respect to the coarse density (inverse-CDF sampling), so round-off of 1e-7 in the coarse pass moves MLP_1's
gradient by 1e-4..1e-3 -- the oracle's own float32 evaluation is 7e-4 / 1e-3 (relative L2) away from its float64 one
(tests/golden/train_grad.npz, tests/test_gpu_trained_state.py).  This script replays the twin through the HIP path with the
gradient of every step multiplied element-wise by (1 + eps * N(0,1)), fresh noise per step, for several eps and noise seeds,
and prints the held-out PSNR each leg ends with: the spread at eps = 1e-3 is the noise floor a HIP-vs-oracle comparison of
this run has; the spread at eps = 1e-6 is what merely reordering a sum does.
    python scripts/twin_noise_sensitivity.py
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

from oracle import nerf_oracle as O  # noqa: E402  (Cfg / init / held-out rays only)
import _helpers as H  # noqa: E402
from _cpu_feeder import feeder_for  # noqa: E402
from plenoctree_amd import ops  # noqa: E402
from plenoctree_amd.nerf_sh.nerf import datasets, models, utils  # noqa: E402


def main():
    datasets.Dataset.feeder_factory = staticmethod(feeder_for)
    dev = torch.device("cuda:0")
    cfg = O.Cfg()
    pcfg = H.pxo_cfg(ops, cfg)
    B, steps = H.TWIN_LONG_RAYS, H.TWIN_LONG_STEPS
    rays, px = H.twin_heldout()
    drays = utils.Rays(*[r.to(dev) for r in rays])
    flat0 = O.flatten_params(O.init_params(cfg, seed=20200823))
    # the run's inputs once, on the device (identical for every leg)
    feed = [(utils.Rays(*[r.to(dev) for r in b["rays"]]), b["pixels"].to(dev), t.to(dev), u.to(dev), sp.to(dev), lr)
            for _, b, t, u, sp, lr in H.twin_steps(B, steps, cfg)]
    legs = [(0.0, 0)] + [(eps, seed) for eps in (1e-6, 1e-4, 1e-3) for seed in (1, 2, 3)]
    if os.environ.get("PXO_TWIN_LEGS") == "short":            # A/B of a variant library: the test's four legs
        legs = [(0.0, 0)] + [(1e-6, seed) for seed in (1, 2, 3)]
    variant = os.environ.get("PXO_LIB", "")
    torch_adam = os.environ.get("PXO_TWIN_ADAM") == "torch"
    for eps, seed in legs:
        model = models.NerfModel(pcfg)
        state = models.TrainState(pcfg, flat0.clone().to(dev))
        gen = torch.Generator(device=dev).manual_seed(1000 + seed)
        ws = state.workspace(ops.train_workspace_bytes(pcfg, B))
        t0 = time.time()
        trace = {}
        for r, pixels, t_rand, u, sp, lr in feed:
            ops.train_fwd_bwd(pcfg, state.params, state.packed, r.origins, r.directions, r.viewdirs, pixels, state.grads,
                              state.stats, ws, randomized=True, t_rand=t_rand, u=u, sp_points=sp)
            if eps > 0:
                state.grads.mul_(1.0 + eps * torch.randn(state.grads.shape, device=dev, generator=gen))
            if torch_adam:
                # component swap: the oracle's Adam (oracle/nerf_oracle.py adam_update, torch float32 ops) on the HIP gradient,
                # then a plain re-pack of the weight images -- does the optimiser kernel carry the twin's late offset?
                g = state.grads
                t = state.step + 1
                state.m.mul_(0.9).add_(g, alpha=1.0 - 0.9)
                state.v.mul_(0.999).add_(g * g, alpha=1.0 - 0.999)
                m_hat = state.m / (1.0 - 0.9 ** t)
                v_hat = state.v / (1.0 - 0.999 ** t)
                state.params.sub_(lr * m_hat / (torch.sqrt(v_hat) + 1e-8))
                state.repack()
            else:
                ops.adam_pack_step(pcfg, state.params, state.m, state.v, state.grads, lr, state.step, state.packed)
            state.step += 1
            if state.step % 250 == 0 and state.step < steps:      # the oracle legs print the same checkpoints
                trace[state.step] = round(H._psnr(model.apply(state, drays, False)[1][0].cpu(), px), 4)
        out = model.apply(state, drays, False)[1][0].cpu()
        print(json.dumps({"library": os.path.basename(variant) or "default", "adam": "torch" if torch_adam else "pxo_adam_pack_step", "grad_noise_eps": eps, "noise_seed": seed, "psnr_heldout": H._psnr(out, px), "trace": trace,
                          "wall_s": round(time.time() - t0, 1)}), flush=True)


if __name__ == "__main__":
    main()
