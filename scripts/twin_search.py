"""HIP leg only of the trained-PSNR twin at candidate (rays, steps) sizes: which horizon leaves the 14 dB regime?
(The oracle leg of the chosen size is then run once by tests/golden/make_trained_twin.py.)
    python scripts/twin_search.py 1024x1500 1024x2000 ...
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

from oracle import nerf_oracle as O  # noqa: E402  (checker-side helpers only: Cfg / init / held-out rays)
import _helpers as H  # noqa: E402
from _cpu_feeder import feeder_for  # noqa: E402
from plenoctree_amd import ops  # noqa: E402
from plenoctree_amd.nerf_sh.nerf import datasets, models, utils  # noqa: E402


def main():
    datasets.Dataset.feeder_factory = staticmethod(feeder_for)
    dev = torch.device("cuda:0")
    cfg = O.Cfg()
    pcfg = H.pxo_cfg(ops, cfg)
    rays, px = H.twin_heldout()
    drays = utils.Rays(*[r.to(dev) for r in rays])
    for spec in sys.argv[1:]:
        B, steps = [int(x) for x in spec.split("x")]
        flat0 = O.flatten_params(O.init_params(cfg, seed=20200823))
        model = models.NerfModel(pcfg)
        state = models.TrainState(pcfg, flat0.clone().to(dev))
        t0 = time.time()
        trace = {}
        for step, batch, t_rand, u, sp, lr in H.twin_steps(B, steps, cfg):
            dbatch = {"rays": utils.Rays(*[r.to(dev) for r in batch["rays"]]), "pixels": batch["pixels"].to(dev)}
            models.train_step(model, state, dbatch, lr, t_rand=t_rand.to(dev), u=u.to(dev), sp_points=sp.to(dev))
            if (step + 1) % 250 == 0:
                out = model.apply(state, drays, False)[1][0].cpu()
                trace[step + 1] = round(H._psnr(out, px), 3)
        out = model.apply(state, drays, False)[1][0].cpu()
        print(json.dumps({"rays": B, "steps": steps, "psnr_heldout": H._psnr(out, px), "trace": trace,
                          "wall_s": round(time.time() - t0, 1)}), flush=True)


if __name__ == "__main__":
    main()
