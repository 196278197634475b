"""Test-set renderer: `python -m plenoctree_amd.nerf_sh.eval --train_dir D --config blender
--data_dir DATA --chunk 4096` (reference: nerf_sh/eval.py:45-133; deterministic sampling,
randomized=False).  Also reports PSNR (the lines the reference keeps commented out, :100-105)."""
import os
import sys

import numpy as np
import torch

from .. import dist
from .nerf import datasets, models, utils


def save_outputs(out_dir, idx, rgb, disp):
    """nerf_sh/eval.py:107-112: the prediction and its disparity map as `<idx>.png` / `disp_<idx>.png`."""
    utils.save_img(rgb, os.path.join(out_dir, "{:03d}.png".format(idx)))
    utils.save_img(disp, os.path.join(out_dir, "disp_{:03d}.png".format(idx)))


def save_summary(out_dir, step, psnrs, ssims):
    """nerf_sh/eval.py:121-129 (the block the reference keeps commented out): mean and per-image PSNR / SSIM."""
    for name, vals in (("psnr", psnrs), ("ssim", ssims)):
        with open(os.path.join(out_dir, f"{name}.txt"), "w") as f:
            f.write("{}".format(np.mean(np.array(vals))))
        with open(os.path.join(out_dir, f"{name}s_{step}.txt"), "w") as f:
            f.write(" ".join([str(v) for v in vals]))


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
    psnrs, ssims = [], []
    for idx in range(0, dataset.size, max(args.approx_eval_skip, 1)):
        ex = dataset.get_image(idx)
        rgb, disp, acc = utils.render_image(lambda r: model.apply(state, r, False), ex["rays"], chunk=args.chunk,
                                            world_size=comm.world, rank=comm.rank, gather=comm.all_gather_cat)
        psnr = utils.compute_psnr(((rgb - ex["pixels"]) ** 2).mean().item())
        psnrs.append(psnr)
        if comm.rank == 0:
            if args.save_output:
                ssim = float(utils.compute_ssim(rgb.clamp(0.0, 1.0), ex["pixels"], max_val=1.0))
                ssims.append(ssim)
                print(f"PSNR = {psnr:.4f}, SSIM = {ssim:.4f}", flush=True)
                save_outputs(out_dir, idx, rgb, disp[..., 0] if disp.dim() == 3 else disp)      # eval.py:110: pred_disp[Ellipsis, 0]
            else:
                print(f"PSNR = {psnr:.4f}", flush=True)
    if comm.rank == 0:
        print(f"Average PSNR {float(np.mean(psnrs)):.4f} over {len(psnrs)} images", flush=True)
        if args.save_output:
            save_summary(out_dir, state.step, psnrs, ssims)
    comm.shutdown()
    return psnrs


if __name__ == "__main__":
    main(sys.argv[1:])
