#!/usr/bin/env python
"""Noise floor of the trained-PSNR comparison (tests/test_gpu_parity.py::test_trained_psnr_matches_oracle_training):
the ORACLE trained in float32 against the ORACLE trained in float64, same 400 Adam steps, same batches and injected
randoms, same held-out rays.  CPU only (test infrastructure: imports oracle/ and tests/_cpu_feeder.py).

    python scripts/trained_psnr_noise_floor.py  ->  tests/golden/trained_psnr_noise_floor.json
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import nerf_oracle as O                      # noqa: E402
from _cpu_feeder import feeder_for                       # noqa: E402
from plenoctree_amd.nerf_sh.nerf import datasets, utils  # noqa: E402


def main():
    datasets.Dataset.feeder_factory = staticmethod(feeder_for)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    cfg = O.Cfg(sparsity_npoints=1000)
    args = utils.define_flags().parse_args(["--config", "synthetic", "--train_dir", "x"])
    utils.update_flags(args); args.factor = 8
    B, steps = 64, 400
    ds = datasets.get_dataset("train", args, torch.device("cpu"), batch_size=B)
    flat0 = O.flatten_params(O.init_params(cfg, seed=20200823))
    st = {torch.float32: [flat0.clone(), torch.zeros_like(flat0), torch.zeros_like(flat0)],
          torch.float64: [flat0.double(), torch.zeros_like(flat0).double(), torch.zeros_like(flat0).double()]}
    for step in range(steps):
        batch = next(ds)
        g = torch.Generator().manual_seed(1000 + step)
        t_rand = torch.rand(B, 64, generator=g); u = torch.rand(B, 128, generator=g)
        sp = (torch.rand(1000, 3, generator=g) * 2 - 1) * 1.5
        lr = utils.learning_rate_decay(step, 5e-4, 5e-6, steps)
        for dt, (p, m, v) in st.items():
            rays = O.Rays(*[r.to(dt) for r in batch["rays"]])
            p, m, v, _, _ = O.train_step(p, m, v, step, rays, batch["pixels"].to(dt), cfg, t_rand.to(dt), u.to(dt), sp.to(dt), lr)
            st[dt] = [p, m, v]
        if step % 50 == 0:
            print("step", step, flush=True)
    test_ds = datasets.get_dataset("test", args, torch.device("cpu"))
    views = [test_ds.get_image(i) for i in (0, 67, 133)]
    rays = O.Rays(*[torch.cat([t["rays"][k].reshape(-1, 3)[::4] for t in views]).contiguous() for k in range(3)])
    px = torch.cat([t["pixels"].reshape(-1, 3)[::4] for t in views])
    psnr = lambda a: float(-10.0 * torch.log10(((a.double() - px.double()) ** 2).mean()))
    with torch.no_grad():
        r32 = O.render(O.unflatten_params(st[torch.float32][0], cfg), rays, cfg)[1][0]
        r64 = O.render(O.unflatten_params(st[torch.float64][0], cfg), O.Rays(*[r.double() for r in rays]), cfg)[1][0]
        cross = O.render(O.unflatten_params(st[torch.float64][0].float(), cfg), rays, cfg)[1][0]   # f64-trained weights, f32 render
    out = {"steps": steps, "rays_per_step": B, "psnr_oracle_f32_trained": psnr(r32), "psnr_oracle_f64_trained": psnr(r64),
           "psnr_oracle_f64_trained_f32_rendered": psnr(cross),
           "noise_floor_db": abs(psnr(r32) - psnr(r64)),
           "param_rel_l2_f32_vs_f64": float((st[torch.float32][0].double() - st[torch.float64][0]).norm() / st[torch.float64][0].norm()),
           "host": f"{os.cpu_count()} cpus, torch {torch.__version__}, threads {torch.get_num_threads()}"}
    path = os.path.join(ROOT, "tests", "golden", "trained_psnr_noise_floor.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
