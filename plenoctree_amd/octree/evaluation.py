"""Octree evaluation on the MI355X path (reference: octree/evaluation.py): loads a tree npz and reports the
mean PSNR of its renders of the test split; test images are sharded over the GPUs.

    python -m plenoctree_amd.octree.evaluation --input tree_opt.npz --config blender --data_dir ... [--write_images DIR]
"""
import os
import sys

import numpy as np
import torch

from .. import dist
from ..nerf_sh.nerf import datasets, utils
from . import extraction
from .svox import N3Tree


def define_flags():
    p = utils.define_flags()
    a = p.add_argument
    a("--input", type=str, default="./tree_opt.npz")            # octree/evaluation.py:54-58
    a("--write_vid", type=str, default=None)                     # :59-63 (mp4 through imageio there; GIF through PIL here)
    a("--write_images", type=str, default=None)                  # :64-68
    a("--renderer_step_size", type=float, default=1e-4)          # octree/nerf/utils.py:211-215
    a("--no_early_stop", action="store_true")
    return p


def main(argv=None):
    args = define_flags().parse_args(argv)
    utils.update_flags(args)
    if args.write_vid is not None and os.path.splitext(args.write_vid)[1].lower() != ".gif":
        # checked BEFORE anything is rendered: imageio / ffmpeg are not installed, so only the animated GIF writer is built
        raise ValueError(f"--write_vid {args.write_vid}: only animated GIF output is built (no imageio/ffmpeg here); "
                         "give a .gif path")
    if not torch.cuda.is_available():
        raise SystemExit("octree.evaluation needs a ROCm GPU; the HIP path has no CPU fallback")
    comm = dist.init_from_env()
    torch.cuda.set_device(comm.local_rank)
    device = torch.device("cuda", comm.local_rank)
    dataset = datasets.get_dataset("test", args, device)
    if comm.rank == 0:
        print("N3Tree load", args.input, flush=True)
    tree = N3Tree.load(args.input, map_location=device)
    want_frames = args.write_images is not None or args.write_vid is not None
    psnr, ssim, frames = extraction.eval_octree(tree, dataset, args, comm, want_frames=want_frames, want_ssim=True)
    if comm.rank == 0:
        print("Average PSNR", psnr, "SSIM", ssim, flush=True)
    if args.write_vid is not None and frames:
        # imageio / ffmpeg are not installed: this rank's frames (every world-th view) as an animated GIF
        from PIL import Image
        path = args.write_vid if comm.world == 1 else os.path.splitext(args.write_vid)[0] + f".rank{comm.rank}.gif"
        print("Writing to", path, flush=True)
        ims = [Image.fromarray(im.numpy()) for _, im in frames]
        ims[0].save(path, save_all=True, append_images=ims[1:], duration=50, loop=0)
    if args.write_images is not None:
        from PIL import Image
        os.makedirs(args.write_images, exist_ok=True)
        for idx, im in frames:
            Image.fromarray(im.numpy()).save(os.path.join(args.write_images, f"{idx:03d}.png"))
    comm.shutdown()
    return psnr


if __name__ == "__main__":
    main(sys.argv[1:])
