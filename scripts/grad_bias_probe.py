#!/usr/bin/env python
"""Is there a SYSTEMATIC component in the HIP gradient's distance from the float64 oracle's?

Single-step parity tests bound the distance (rel L2 ~1e-3 for MLP_1: the float32 noise floor of this function, set by the
ill-conditioned inverse-CDF sampling); a bias far below that bound would pass them and still steer 1,500 training steps.  This
probe evaluates K batches of the long twin AT the oracle-trained parameters (tests/golden/trained_twin_1024x1500.npz: the late,
low-learning-rate regime) three ways -- HIP, oracle float32, oracle float64 -- and looks at the MEAN error over the batches:
noise averages out like 1/sqrt(K), a bias does not.  Printed per MLP: the per-batch error, the error of the batch-mean gradient
for HIP and for the float32 oracle, and the cosine of each mean error with the mean gradient (a component along the gradient is
a step-size error, the kind that would move a converged PSNR).
    python scripts/grad_bias_probe.py [K]
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

from oracle import nerf_oracle as O  # noqa: E402
import _helpers as H  # noqa: E402
from _cpu_feeder import feeder_for  # noqa: E402
from plenoctree_amd import ops  # noqa: E402
from plenoctree_amd.nerf_sh.nerf import datasets  # noqa: E402


def main():
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    datasets.Dataset.feeder_factory = staticmethod(feeder_for)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    dev = torch.device("cuda:0")
    cfg = O.Cfg()
    pcfg = H.pxo_cfg(ops, cfg)
    B = H.TWIN_LONG_RAYS
    g = np.load(os.path.join(ROOT, "tests", "golden", f"trained_twin_{B}x{H.TWIN_LONG_STEPS}.npz"))
    flat = torch.tensor(g["params"])
    n = flat.numel() // 2
    fd = flat.to(dev)
    packed = [ops.pack_weights(pcfg, fd[i * n:(i + 1) * n].contiguous()) for i in range(2)]
    ws = torch.empty(ops.train_workspace_bytes(pcfg, B), dtype=torch.uint8, device=dev)
    G = {"hip": [], "f32": [], "f64": []}
    t0 = time.time()
    for step, batch, t_rand, u, sp, _ in H.twin_steps(B, K, cfg):
        rays = O.Rays(*batch["rays"]); px = batch["pixels"]
        grads = torch.zeros_like(fd); stats = torch.zeros(6, device=dev)
        ops.train_fwd_bwd(pcfg, fd, packed, *[r.to(dev) for r in rays], px.to(dev), grads, stats, ws, randomized=True,
                          t_rand=t_rand.to(dev), u=u.to(dev), sp_points=sp.to(dev))
        G["hip"].append(grads.cpu().double())
        G["f32"].append(H.oracle_loss_and_grad_chunked(flat, rays, px, cfg, t_rand, u, sp, torch.float32)[1].double())
        G["f64"].append(H.oracle_loss_and_grad_chunked(flat, rays, px, cfg, t_rand, u, sp, torch.float64)[1])
    G = {k: torch.stack(v) for k, v in G.items()}
    out = {"batches": K, "rays": B, "wall_s": round(time.time() - t0, 1)}
    for mi, (lo, hi) in enumerate(((0, n), (n, 2 * n))):
        ref = G["f64"][:, lo:hi]
        mean_ref = ref.mean(0)
        for who in ("hip", "f32"):
            err = G[who][:, lo:hi] - ref
            per_batch = float((err.norm(dim=1) / ref.norm(dim=1)).mean())
            mean_err = err.mean(0)
            out[f"mlp{mi}_{who}"] = {
                "per_batch_rel_err": per_batch,
                "rel_err_of_mean_gradient": float(mean_err.norm() / mean_ref.norm()),
                "expected_if_pure_noise": per_batch * float(ref.norm(dim=1).mean() / mean_ref.norm()) / K ** 0.5,
                "cos_mean_err_with_mean_gradient": float((mean_err @ mean_ref) / (mean_err.norm() * mean_ref.norm())),
            }
    print(json.dumps(out))


if __name__ == "__main__":
    main()
