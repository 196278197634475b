"""Test-set renderer: `python -m plenoctree_amd.nerf_sh.eval --train_dir D --config blender
--data_dir DATA --chunk 4096` (reference: nerf_sh/eval.py:45-133; deterministic sampling,
randomized=False).  Also reports PSNR (the lines the reference keeps commented out, :100-105)."""
import os
import sys

import numpy as np
import torch

from .. import dist
from .nerf import datasets, models, utils


def main(argv=None):
    args = utils.define_flags().parse_args(argv)
    utils.update_flags(args)
    if not torch.cuda.is_available():
        raise SystemExit("nerf_sh.eval needs a ROCm GPU; the HIP path has no CPU fallback")
    comm = dist.init_from_env()
    torch.cuda.set_device(comm.local_rank)
    device = torch.device("cuda", comm.local_rank)
    utils.check_flags(args, world_size=comm.world)
    dataset = datasets.get_dataset("test", args, device)
    model, state = models.get_model_state(args, device, restore=True)
    out_dir = os.path.join(args.train_dir, "test_preds")
    if args.save_output and comm.rank == 0:
        os.makedirs(out_dir, exist_ok=True)
    psnrs = []
    for idx in range(0, dataset.size, max(args.approx_eval_skip, 1)):
        ex = dataset.get_image(idx)
        rgb, disp, acc = utils.render_image(lambda r: model.apply(state, r, False), ex["rays"], chunk=args.chunk,
                                            world_size=comm.world, rank=comm.rank, gather=comm.all_gather_cat)
        psnr = utils.compute_psnr(((rgb - ex["pixels"]) ** 2).mean().item())
        psnrs.append(psnr)
        if comm.rank == 0:
            print(f"PSNR = {psnr:.4f}", flush=True)
            if args.save_output:
                from PIL import Image
                Image.fromarray((np.clip(rgb.cpu().numpy(), 0, 1) * 255).astype(np.uint8)).save(
                    os.path.join(out_dir, f"{idx:03d}.png"))
    if comm.rank == 0:
        print(f"Average PSNR {float(np.mean(psnrs)):.4f} over {len(psnrs)} images", flush=True)
    comm.shutdown()
    return psnrs


if __name__ == "__main__":
    main(sys.argv[1:])
