"""Mesh of a trained NeRF-SH's sigma isosurface (reference: nerf_sh/gen_mesh.py): sigma on a dense grid through the
HIP point evaluator (the sigma-only head pass, same kernel as the extraction grid), marching cubes on the host, OBJ out.

    python -m plenoctree_amd.nerf_sh.gen_mesh --train_dir D --config blender --reso "300 300 300" --iso 6.0

PyMCubes is not installed here; `isosurface.marching_cubes` takes its place (same vertices, see that module).
"""
import os
import sys

import numpy as np
import torch

from . import isosurface
from .nerf import models, utils


def define_flags():
    """nerf_sh/gen_mesh.py:49-76."""
    p = utils.define_flags()
    a = p.add_argument
    a("--reso", type=str, default="300 300 300", help="marching cubes resolution in each dimension: x y z")
    a("--c1", type=str, default="-2 -2 -2", help="lower corner, x y z or one number")
    a("--c2", type=str, default="2 2 2", help="upper corner, x y z or one number")
    a("--iso", type=float, default=6.0, help="sigma isosurface")
    a("--coarse", type=utils._bool, nargs="?", const=True, default=False, help="force the coarse network")
    a("--point_chunk", type=int, default=720720, help="points per evaluator launch (--chunk is ignored)")
    return p


def sigma_grid(fn, c1, c2, reso, chunk, device):
    """:105-119: raw sigma at linspace(c1, c2, reso) per axis (end points included), ij order -> float32 [rx, ry, rz].
    `fn(points [n,3]) -> raw_sigma [n,1]`.  The axis coordinates are numpy float32 linspaces like the reference's; the
    point list is assembled per chunk on the device instead of materialising the [N,3] host array."""
    axes = [torch.from_numpy(np.linspace(lo, hi, sz, dtype=np.float32)).to(device) for lo, hi, sz in zip(c1, c2, reso)]
    ny, nz = reso[1], reso[2]
    total = reso[0] * ny * nz
    out = torch.empty(total, dtype=torch.float32, device=device)
    for s in range(0, total, chunk):
        idx = torch.arange(s, min(s + chunk, total), device=device)
        pts = torch.stack([axes[0][idx // (ny * nz)], axes[1][(idx // nz) % ny], axes[2][idx % nz]], 1)
        out[s:s + idx.numel()] = fn(pts.contiguous()).reshape(-1)
    return out.reshape(*reso)


def marching_cubes(fn, c1, c2, reso, isosurface_level, chunk, device):
    """:88-130.  Vertices are scaled by (c2 - c1) / reso exactly as the reference does (:127) - note the samples sit
    at spacing (c2 - c1) / (reso - 1), so the reference's mesh is shrunk by (reso-1)/reso towards c1; kept for parity."""
    sigmas = sigma_grid(fn, c1, c2, reso, chunk, device).cpu().numpy()
    print("* Running marching cubes", flush=True)
    vertices, triangles = isosurface.marching_cubes(sigmas, isosurface_level)
    c1, c2 = np.array(c1), np.array(c2)
    vertices = vertices * ((c2 - c1) / np.array(reso))
    return vertices + c1, triangles


def save_obj(vertices, triangles, path, vert_rgb=None):
    """:133-158: `v x y z [r g b]` with four decimals, then 1-based `f a b c`."""
    with open(path, "w") as f:
        if vert_rgb is None:
            f.writelines("v %.4f %.4f %.4f\n" % tuple(v) for v in vertices)
        else:
            f.writelines("v %.4f %.4f %.4f %.4f %.4f %.4f\n" % (*v, *c) for v, c in zip(vertices, vert_rgb))
        f.writelines("f %d %d %d\n" % tuple(t) for t in (np.asarray(triangles) + 1))


def _triple(s, cast):
    v = [cast(x) for x in s.split()]
    return v * 3 if len(v) == 1 else v


def main(argv=None):
    """:161-194."""
    args = define_flags().parse_args(argv)
    utils.update_flags(args)
    if not torch.cuda.is_available():
        raise SystemExit("nerf_sh.gen_mesh needs a ROCm GPU; the HIP path has no CPU fallback")
    device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0)))
    torch.cuda.set_device(device)
    utils.check_flags(args, require_data=False)
    reso, c1, c2 = _triple(args.reso, int), _triple(args.c1, float), _triple(args.c2, float)
    print("* Creating model", flush=True)
    model, state = models.get_model_state(args, device, restore=True)
    print("* Eval reso", args.reso, "coarse?", args.coarse, flush=True)
    print("* Evaluating sigma @", reso[0] * reso[1] * reso[2], "points", flush=True)

    def fn(points):
        return model.eval_points_raw(state, points, coarse=args.coarse, want_rgb=False)[1]

    verts, faces = marching_cubes(fn, c1, c2, reso, args.iso, args.point_chunk, device)
    mesh_path = os.path.join(args.train_dir, "mesh.obj")
    print(" Saving to", mesh_path, flush=True)
    save_obj(verts, faces, mesh_path)
    return verts, faces


if __name__ == "__main__":
    main(sys.argv[1:])
