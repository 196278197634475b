"""Turntable renderer of a trained NeRF-SH (reference: nerf_sh/gen_video.py): poses on a circle at the given
elevation -> rays -> deterministic render -> PNG frames under <train_dir>/video/e<elev>/frames (+ an animated GIF;
the reference writes an mp4 through imageio, which is not installed here).

    python -m plenoctree_amd.nerf_sh.gen_video --train_dir D --config blender --num_views 40 [--write_poses poses.txt]
"""
import os
import sys

import numpy as np
import torch

from .. import dist, ops
from .nerf import models, utils


def define_flags():
    """nerf_sh/gen_video.py:52-105."""
    p = utils.define_flags()
    a = p.add_argument
    a("--elevation", type=float, default=-30.0)
    a("--num_views", type=int, default=40)
    a("--height", type=int, default=800)
    a("--width", type=int, default=800)
    a("--camera_angle_x", "-A", type=float, default=0.7)
    a("--intrin", type=str, default=None)
    a("--radius", type=float, default=4.0)
    a("--fps", type=int, default=20)
    a("--up_axis", type=int, default=1)
    a("--write_poses", type=str, default=None)
    return p


def render_poses(args):
    """:113-120: num_views poses on a circle, angles linspace(-180, 180, n+1)[:-1]."""
    return np.stack([utils.pose_spherical(angle, args.elevation, args.radius, args.up_axis - 1)
                     for angle in np.linspace(-180, 180, args.num_views + 1)[:-1]], 0)


def main(argv=None):
    args = define_flags().parse_args(argv)
    utils.update_flags(args)
    if not torch.cuda.is_available():
        raise SystemExit("nerf_sh.gen_video needs a ROCm GPU; the HIP path has no CPU fallback")
    comm = dist.init_from_env()
    torch.cuda.set_device(comm.local_rank)
    device = torch.device("cuda", comm.local_rank)
    utils.check_flags(args, require_data=False, world_size=comm.world)
    say = print if comm.rank == 0 else (lambda *a, **k: None)
    say("* Generating poses", flush=True)
    poses = render_poses(args)
    if args.write_poses and comm.rank == 0:
        np.savetxt(args.write_poses, poses.reshape(-1, 4))
        print("Saved poses to", args.write_poses, flush=True)
    focal = 0.5 * args.width / np.tan(0.5 * args.camera_angle_x)
    if args.intrin is not None:
        K = np.loadtxt(args.intrin)
        focal = (K[0, 0] + K[1, 1]) * 0.5
    say("* Creating model", flush=True)
    model, state = models.get_model_state(args, device, restore=True)
    video_dir = os.path.join(args.train_dir, "video", "e{:03}".format(int(-args.elevation * 10)))
    frames_dir = os.path.join(video_dir, "frames")
    say(" Saving to", video_dir, flush=True)
    if comm.rank == 0:
        os.makedirs(frames_dir, exist_ok=True)
    c2w = torch.from_numpy(np.ascontiguousarray(poses[:, :3, :4])).to(device)
    frames = []
    for i in range(args.num_views):
        say(f"** View {i + 1}/{args.num_views} = {i / args.num_views * 100}%", flush=True)
        rays = utils.Rays(*[r.reshape(args.height, args.width, 3)
                            for r in ops.generate_rays(c2w[i], args.width, args.height, focal)])
        rgb, disp, acc = utils.render_image(lambda r: model.apply(state, r, False), rays, chunk=args.chunk,
                                            world_size=comm.world, rank=comm.rank, gather=comm.all_gather_cat)
        if comm.rank == 0:
            utils.save_img(rgb, os.path.join(frames_dir, f"{i:04}.png"))
            frames.append((np.clip(rgb.cpu().numpy(), 0.0, 1.0) * 255).astype(np.uint8))
    if comm.rank == 0 and frames:
        from PIL import Image
        vid_path = os.path.join(video_dir, "video.gif")
        print("* Writing", vid_path, flush=True)
        ims = [Image.fromarray(f) for f in frames]
        ims[0].save(vid_path, save_all=True, append_images=ims[1:], duration=int(1000 / max(args.fps, 1)), loop=0)
        print("* Done", flush=True)
    comm.shutdown()
    return frames


if __name__ == "__main__":
    main(sys.argv[1:])
