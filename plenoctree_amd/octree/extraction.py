"""Dense-grid evaluation stage of NeRF-SH -> PlenOctree extraction on the MI355X path.

Covers the MLP loops of the reference's octree/extraction.py: `auto_scale` (:244-286), step 1's
grid sigma evaluation (:288-320) and step 2's per-leaf sample evaluation + mean (:355-394).  The
octree itself (svox N3Tree build/refine/sample/assign, grid_weight_render, npz format) is svox code
outside this tree and is not reimplemented here: `main` stops at the sigma grid /
sigma mask, which is exactly what the tree-building stage consumes.

    python -m plenoctree_amd.octree.extraction --train_dir D --config blender --output grid.npz
    python -m torch.distributed.run --nproc-per-node 8 -m plenoctree_amd.octree.extraction ...
"""
import sys
import time

import numpy as np
import torch

from .. import dist, ops
from ..nerf_sh.nerf import models, utils


def tree_transform(center, radius):
    """N3Tree's offset / invradius (octree/extraction.py:250-251)."""
    radius = np.broadcast_to(np.asarray(radius, np.float32), (3,)).copy()
    center = np.broadcast_to(np.asarray(center, np.float32), (3,)).copy()
    return 0.5 * (1.0 - center / radius), 0.5 / radius


def grid_sigma(model, state, reso, center, radius, comm=None):
    """sigma of MLP_1 (fine) on the reso^3 grid, x slowest (extraction.py:294-320), evaluated in
    x-slabs sharded over the ranks of `comm` and all-gathered (every rank returns the full grid)."""
    comm = comm or dist.Comm()
    offset, scale = tree_transform(center, radius)
    which = 1 if model.num_fine_samples > 0 else 0
    base, rem = divmod(reso, comm.world)
    width = base + (1 if rem else 0)                      # equal-size buffers for all_gather
    x0, x1 = dist.slab_range(reso, comm.world, comm.rank)
    slab = torch.zeros(width * reso * reso, dtype=torch.float32, device=state.params.device)
    if x1 > x0:
        ops.grid_sigma(model.cfg, state.packed[which][0], reso, x0, x1, offset, scale,
                       out=slab[: (x1 - x0) * reso * reso])
    if not comm.is_dist:
        return slab[: reso ** 3]
    full = comm.all_gather_cat(slab).reshape(comm.world, width * reso * reso)
    parts = []
    for r in range(comm.world):
        a, b = dist.slab_range(reso, comm.world, r)
        parts.append(full[r, : (b - a) * reso * reso])
    return torch.cat(parts)


def auto_scale(model, state, center, radius, init_grid_depth=8, scale_alpha_thresh=0.01, comm=None):
    """Bounding box of sigma >= thresh on a 2^depth grid (extraction.py:244-286)."""
    reso = 2 ** init_grid_depth
    sig = grid_sigma(model, state, reso, center, radius, comm)
    sigma_thresh = -np.log(1.0 - scale_alpha_thresh) / (2.0 / reso)
    offset, scale = tree_transform(center, radius)
    mask = (sig >= sigma_thresh).reshape(reso, reso, reso)
    arr = (torch.arange(reso, dtype=torch.float32, device=sig.device) + 0.5) / reso
    lc, uc = [], []
    for ax in range(3):
        other = tuple(a for a in range(3) if a != ax)
        occ = mask.any(dim=other)
        if not bool(occ.any()):
            raise RuntimeError("auto_scale: no voxel above the sigma threshold")
        coords = ((arr - float(offset[ax])) / float(scale[ax]))[occ]
        lc.append(float(coords.min()) - 0.5 / reso)
        uc.append(float(coords.max()) + 0.5 / reso)
    lc, uc = np.array(lc), np.array(uc)
    return ((lc + uc) * 0.5).tolist(), ((uc - lc) * 0.5).tolist()


def eval_leaf_samples(model, state, points, samples_per_cell):
    """Step 2 (extraction.py:367-393, SH/SG formats): [n_cells*S, 3] points -> mean over the S
    samples of cat([raw_rgb, raw_sigma]) -> [n_cells, 3K+1]."""
    rgb, sigma = model.eval_points_raw(state, points)
    return ops.mean_over_samples(model.cfg, rgb, sigma, samples_per_cell)


def main(argv=None):
    p = utils.define_flags()
    p.add_argument("--output", type=str, default=None, help="npz with the sigma grid / mask")
    p.add_argument("--center", type=float, nargs=3, default=[0.0, 0.0, 0.0])
    p.add_argument("--radius", type=float, nargs=3, default=[1.5, 1.5, 1.5])     # extraction.py:76-79
    p.add_argument("--init_grid_depth", type=int, default=8)
    p.add_argument("--alpha_thresh", type=float, default=0.01)
    p.add_argument("--scale_alpha_thresh", type=float, default=0.01)
    p.add_argument("--autoscale", action="store_true")
    args = p.parse_args(argv)
    utils.update_flags(args)
    if not torch.cuda.is_available():
        raise SystemExit("octree.extraction needs a ROCm GPU; the HIP path has no CPU fallback")
    comm = dist.init_from_env()
    torch.cuda.set_device(comm.local_rank)
    device = torch.device("cuda", comm.local_rank)
    utils.check_flags(args, require_data=False, world_size=comm.world)
    model, state = models.get_model_state(args, device, restore=True)
    center, radius = args.center, args.radius
    if args.autoscale:
        center, radius = auto_scale(model, state, center, radius, args.init_grid_depth, args.scale_alpha_thresh, comm)
        if comm.rank == 0:
            print("* Auto scale result center", center, "radius", radius, flush=True)
    reso = 2 ** (args.init_grid_depth + 1)
    if comm.rank == 0:
        print("* Step 1: Grid eval", reso, flush=True)
    torch.cuda.synchronize(); comm.barrier()
    t0 = time.time()
    sig = grid_sigma(model, state, reso, center, radius, comm)
    torch.cuda.synchronize(); comm.barrier()
    dt = time.time() - t0
    if comm.rank == 0:
        flop = reso ** 3 * (1007104 if model.sh_deg == 3 else 1020928)
        print(f"* grid eval: {reso ** 3} points in {dt:.3f} s = {reso ** 3 / dt / 1e6:.1f} Mpts/s "
              f"({flop / dt / 1e12:.1f} TFLOP/s over {comm.world} GPU(s))", flush=True)
    sigma_thresh = -np.log(1.0 - args.alpha_thresh) / (2.0 / reso)
    mask = sig >= sigma_thresh
    if comm.rank == 0:
        print(f"* {int(mask.sum())} / {reso ** 3} voxels above sigma threshold {sigma_thresh:.4f}", flush=True)
        if args.output:
            np.savez(args.output, sigma=sig.cpu().numpy().reshape(reso, reso, reso), center=np.array(center),
                     radius=np.array(radius), sigma_thresh=sigma_thresh)
    comm.shutdown()
    return sig


if __name__ == "__main__":
    main(sys.argv[1:])
