"""NeRF-SH -> PlenOctree extraction on the MI355X path (reference: octree/extraction.py).

  step 0  auto_scale (:244-286)            sigma on a 2^depth grid -> bounding box
  step 1  grid eval (:288-320)             sigma of MLP_1 on the 2^(depth+1) grid, x-slabs sharded over the GPUs
          masking (:322-335)               "sigma": sigma >= thresh; "weight": max compositing weight over the
                                           training views (calculate_grid_weights :181-214), cameras sharded
          tree build (:337-352)            N3Tree.refine_from_mask (Morton pyramid + scans)
  step 2  3-D antialiasing (:355-394)      S samples per deepest-level cell -> MLP_1 -> mean -> tree data,
                                           nodes sharded over the GPUs
  finish  relu on sigma, save npz, evaluate (:503-516)

    python -m plenoctree_amd.octree.extraction --train_dir D --config blender --data_dir ... --output tree.npz
    python -m torch.distributed.run --nproc-per-node 8 -m plenoctree_amd.octree.extraction ...
"""
import os
import sys
import time

import numpy as np
import torch

from .. import dist, ops
from .. import octree_ops as oops
from ..nerf_sh.nerf import datasets, models, utils
from .svox import N3Tree, VolumeRenderer


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
    x0, x1 = dist.slab_range(reso, comm.world, comm.rank)
    dev = state.params.device
    plane = reso * reso
    if reso % comm.world == 0:
        # equal slabs (the 8-way split of the 512^3 grid): every rank evaluates straight into its rows of the final
        # buffer and the all-gather lands in place -- no padded staging buffer, no second 537 MB concatenation
        full = torch.empty(reso * plane, dtype=torch.float32, device=dev)
        mine = full[x0 * plane: x1 * plane]
        ops.grid_sigma(model.cfg, state.packed[which][0], reso, x0, x1, offset, scale, out=mine)
        if comm.is_dist:
            comm.all_gather_into(full, mine)
        return full
    base, rem = divmod(reso, comm.world)
    width = base + 1                                      # ragged slabs: equal-size padded buffers for all_gather
    slab = torch.zeros(width * plane, dtype=torch.float32, device=dev)
    if x1 > x0:
        ops.grid_sigma(model.cfg, state.packed[which][0], reso, x0, x1, offset, scale, out=slab[: (x1 - x0) * plane])
    full = comm.all_gather_cat(slab).reshape(comm.world, width * plane)
    parts = []
    for r in range(comm.world):
        a, b = dist.slab_range(reso, comm.world, r)
        parts.append(full[r, : (b - a) * plane])
    return torch.cat(parts)


def auto_scale(model, state, center, radius, init_grid_depth=8, scale_alpha_thresh=0.01, comm=None, z_min=None,
               z_max=None):
    """Bounding box of sigma >= thresh on a 2^depth grid (extraction.py:244-286)."""
    reso = 2 ** init_grid_depth
    sig = grid_sigma(model, state, reso, center, radius, comm)
    sigma_thresh = -np.log(1.0 - scale_alpha_thresh) / (2.0 / reso)
    offset, scale = tree_transform(center, radius)
    mask = z_range_mask((sig >= sigma_thresh), reso, offset, scale, z_min, z_max).reshape(reso, reso, reso)
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


def calculate_grid_weights(dataset, sigmas, reso, invradius, offset, step_size, comm=None):
    """calculate_grid_weights (:181-214): per-voxel maximum over the training cameras, camera-sharded."""
    comm = comm or dist.Comm()
    opts = oops.render_opts(step_size=step_size, sigma_thresh=0.0, stop_thresh=0.0)
    cams = torch.from_numpy(np.ascontiguousarray(dataset.camtoworlds[comm.rank::comm.world, :3, :4])).to(sigmas.device)
    weight = torch.zeros(reso ** 3, dtype=torch.float32, device=sigmas.device)
    if cams.shape[0]:
        oops.grid_weight_render(sigmas, reso, cams, dataset.focal, dataset.focal, dataset.w, dataset.h, opts, offset,
                                invradius, grid_weight=weight)
    return comm.all_reduce_max(weight)


def z_range_mask(mask, reso, offset, scale, z_min, z_max):
    """The reference drops grid planes with world z outside [z_min, z_max] before it builds the point list
    (:257-260, :298-301; NDC scenes); on the dense mask (x slowest, z fastest) that is clearing those planes."""
    if z_min is None and z_max is None:
        return mask
    arr = (torch.arange(reso, dtype=torch.float32, device=mask.device) + 0.5) / reso
    zz = (arr - float(offset[2])) / float(scale[2])
    keep = torch.ones(reso, dtype=torch.bool, device=mask.device)
    if z_min is not None:
        keep &= zz >= z_min
    if z_max is not None:
        keep &= zz <= z_max
    m = mask.view(reso, reso, reso)
    m &= keep.to(m.dtype).view(1, 1, reso) if m.dtype != torch.bool else keep.view(1, 1, reso)
    return mask


def step1(args, tree, model, state, dataset, comm):
    """Grid evaluation, masking and tree build (:288-352)."""
    reso = 2 ** (args.init_grid_depth + 1)
    center, radius = tree_center_radius(tree)
    sig = grid_sigma(model, state, reso, center, radius, comm)
    approx_delta = 2.0 / reso
    if args.masking_mode == "sigma":
        mask = oops.threshold_mask(sig, -np.log(1.0 - args.alpha_thresh) / approx_delta)
    elif args.masking_mode == "weight":
        weights = calculate_grid_weights(dataset, sig, reso, tree.invradius, tree.offset, args.renderer_step_size, comm)
        mask = oops.threshold_mask(weights, args.weight_thresh)
        del weights
    else:
        raise ValueError(args.masking_mode)
    del sig
    offset, scale = tree_transform(center, radius)
    mask = z_range_mask(mask, reso, offset, scale, getattr(args, "z_min", None), getattr(args, "z_max", None))
    tree.refine_from_mask(mask)
    if tree.max_depth != args.init_grid_depth:
        raise RuntimeError(f"empty mask: the tree has depth {tree.max_depth}, expected {args.init_grid_depth} "
                           "(lower --weight_thresh / --alpha_thresh or check the bounding box)")
    return mask


def step2(args, tree, model, state, comm, seed=0):
    """3-D antialiasing (:355-394, SH formats): every deepest-level cell gets the mean of the network output
    at `samples_per_cell` uniform points inside it.  Nodes are sharded over the ranks and all-gathered."""
    S = args.samples_per_cell
    _, total = tree.max_depth_nodes()
    per = (total + comm.world - 1) // comm.world
    a, b = min(comm.rank * per, total), min((comm.rank + 1) * per, total)
    chunk_nodes = max(args.chunk // (8 * S), 1) * 64
    block = torch.zeros(per * 8, tree.data_dim, dtype=torch.float32, device=tree.device)
    for n0 in range(a, b, chunk_nodes):
        cnt = min(chunk_nodes, b - n0)
        pts = tree.sample_max_depth_cells(S, first=n0, count=cnt, seed=seed)
        rgb, sigma = model.eval_points_raw(state, pts.view(-1, 3))
        ops.mean_over_samples(model.cfg, rgb, sigma, S, out=block[(n0 - a) * 8:(n0 - a + cnt) * 8])
    if comm.is_dist:
        block = comm.all_gather_cat(block)
    tree.max_depth_data().copy_(block[: total * 8])


def tree_center_radius(tree):
    radius = 0.5 / tree.invradius
    center = (1.0 - 2.0 * tree.offset) * radius
    return center.tolist(), radius.tolist()


@torch.no_grad()
def eval_octree(tree, dataset, args, comm=None, want_frames=False, want_ssim=False):
    """eval_octree (octree/nerf/utils.py:448-497): mean PSNR (and SSIM if asked) of the octree renders of a split;
    images are sharded over the ranks.  LPIPS needs pretrained VGG weights, which cannot be fetched here."""
    comm = comm or dist.Comm()
    r = VolumeRenderer(tree, step_size=args.renderer_step_size)
    acc = torch.zeros(3, dtype=torch.float64, device=tree.device)
    frames = []
    for idx in range(comm.rank, dataset.size, comm.world):
        gt = dataset.get_image(idx)["pixels"]
        im = r.render_persp(torch.from_numpy(dataset.camtoworlds[idx]), width=dataset.w, height=dataset.h,
                            fx=dataset.focal, fast=not args.no_early_stop)
        sse, _ = oops.image_mse(im, gt.contiguous(), want_grad=False)
        acc[0] += utils.compute_psnr(float(sse) / im.numel())
        acc[1] += 1
        if want_ssim:
            acc[2] += float(utils.compute_ssim(im.clamp(0.0, 1.0), gt, max_val=1.0))
        if want_frames:
            frames.append((idx, (im.clamp(0, 1) * 255).to(torch.uint8).cpu()))   # 1.9 MB per 800x800 view
    comm.all_reduce_sum(acc)
    if want_ssim:
        return float(acc[0] / acc[1]), float(acc[2] / acc[1]), frames
    return float(acc[0] / acc[1]), frames


def _floats(text, name):
    vals = [float(v) for v in str(text).split()]
    if len(vals) == 1:
        vals *= 3
    if len(vals) != 3:
        raise ValueError(f"--{name} takes one or three numbers")
    return vals


def define_flags():
    """Flag names and defaults of octree/extraction.py:43-176 and octree/nerf/utils.py:211-219."""
    p = utils.define_flags()
    a = p.add_argument
    a("--output", type=str, default="./tree.npz")
    a("--center", type=str, default="0 0 0")
    a("--radius", type=str, default="1.5")
    a("--alpha_thresh", type=float, default=0.01)
    a("--init_grid_depth", type=int, default=8)
    a("--samples_per_cell", "-S", type=int, default=8)
    a("--masking_mode", type=str, default="weight", choices=["sigma", "weight"])
    a("--weight_thresh", type=float, default=0.001)
    a("--bbox_from_data", action="store_true")
    a("--data_bbox_scale", type=float, default=1.0)
    a("--autoscale", action="store_true")
    a("--bbox_cube", action="store_true")
    a("--bbox_scale", type=float, default=1.0)
    a("--scale_alpha_thresh", type=float, default=0.01)
    a("--tree_branch_n", type=int, default=2)
    a("--eval", type=utils._bool, default=True)
    a("--max_refine_prop", type=float, default=0.5)          # defined by the reference (:86-90), read nowhere
    a("--z_min", type=float, default=None)                   # :91-100: keep only grid points with z_min <= z <= z_max
    a("--z_max", type=float, default=None)
    a("--is_jaxnerf_ckpt", type=utils._bool, nargs="?", const=True, default=False)   # :117-121; see main()
    # not a reference flag: a `*.ckpt` holding more than tensors (argparse.Namespace, numpy scalars ...) is unpickled in full, like
    # the reference's plain torch.load, only when this says so (the default reads tensors only and explains itself otherwise)
    a("--trust_ckpt_pickle", type=utils._bool, nargs="?", const=True, default=False)
    a("--projection_samples", type=int, default=10000)       # :133-137: SH projection of a view-dependent NeRF only
    a("--renderer_step_size", type=float, default=1e-4)
    a("--no_early_stop", action="store_true")
    return p


def load_nerf_checkpoint(args, state):
    """get_model_state(args, restore=True) of octree/nerf/models.py:38-49.  The reference reads one of two formats, chosen by
    --is_jaxnerf_ckpt (:45): a flax-msgpack `checkpoint_<step>` of nerf_sh.train (restore_model_state_from_jaxnerf, :66-113) or
    a torch state dict `*.ckpt` of its torch twin (restore_model_state, :52-63).  Both are read here.  Where the reference, with
    no file of the chosen format in train_dir, silently goes on with the freshly initialised network, this raises -- except that
    without the flag a train_dir holding only flax checkpoints (what nerf_sh.train here and the reference's JAX trainer write)
    is read as such, and said so."""
    from ..nerf_sh.nerf import checkpoints
    if args.is_jaxnerf_ckpt:
        path = checkpoints.restore_checkpoint(args.train_dir, state)
        if path is None:
            raise FileNotFoundError(f"--is_jaxnerf_ckpt: no flax checkpoint_<step> in {args.train_dir}")
        return f"* restore ckpt from {path} (flax msgpack)"
    path = checkpoints.restore_torch_checkpoint(args.train_dir, state, trust_pickle=bool(getattr(args, "trust_ckpt_pickle", False)))
    if path is not None:
        return f"* restore ckpt from {path}. (torch state dict)"
    path = checkpoints.restore_checkpoint(args.train_dir, state)
    if path is None:
        raise FileNotFoundError(f"no *.ckpt (torch state dict) and no checkpoint_<step> (flax msgpack) in {args.train_dir}: "
                                "nothing to extract from")
    return f"* no *.ckpt in {args.train_dir}: restore ckpt from {path} (flax msgpack, as with --is_jaxnerf_ckpt)"


def main(argv=None):
    args = define_flags().parse_args(argv)
    utils.update_flags(args)
    if not torch.cuda.is_available():
        raise SystemExit("octree.extraction needs a ROCm GPU; the HIP path has no CPU fallback")
    comm = dist.init_from_env()
    torch.cuda.set_device(comm.local_rank)
    device = torch.device("cuda", comm.local_rank)
    utils.check_flags(args, require_data=True, world_size=comm.world)
    say = print if comm.rank == 0 else (lambda *a, **k: None)
    say("* Loading NeRF", flush=True)
    model, state = models.get_model_state(args, device, restore=False)
    say(load_nerf_checkpoint(args, state), flush=True)
    dataset = datasets.get_dataset("train", args, device)
    if args.bbox_from_data:                                  # :447-451 (NSVF datasets carry bbox.txt)
        bbox = getattr(dataset, "bbox", None)
        if bbox is None:
            raise ValueError("--bbox_from_data needs a dataset with bbox.txt (NSVF format)")
        center = ((bbox[:3] + bbox[3:6]) * 0.5).tolist()
        radius = ((bbox[3:6] - bbox[:3]) * 0.5 * args.data_bbox_scale).tolist()
        say("Bounding box from data: c", center, "r", radius, flush=True)
    else:
        center, radius = _floats(args.center, "center"), _floats(args.radius, "radius")
    if args.autoscale:
        say("* Step 0: Auto scale", flush=True)
        center, radius = auto_scale(model, state, center, radius, args.init_grid_depth, args.scale_alpha_thresh, comm,
                                    args.z_min, args.z_max)
        say("Autoscale result center", center, "radius", radius, flush=True)
    radius = [r * args.bbox_scale for r in radius]
    if args.bbox_cube:
        radius = [max(radius)] * 3
    data_dim = 1 + args.num_rgb_channels * (args.sh_deg + 1) ** 2
    say("data dim is", data_dim, flush=True)
    tree = N3Tree(N=args.tree_branch_n, data_dim=data_dim, init_refine=0, depth_limit=args.init_grid_depth,
                  radius=radius, center=center, data_format=f"SH{(args.sh_deg + 1) ** 2}", map_location=device)
    reso = 2 ** (args.init_grid_depth + 1)
    say("* Step 1: Grid eval", reso, flush=True)
    torch.cuda.synchronize(); comm.barrier(); t0 = time.time()
    mask = step1(args, tree, model, state, dataset, comm)
    torch.cuda.synchronize(); comm.barrier(); t1 = time.time()
    say(f"  {int(mask.sum())} / {reso ** 3} voxels kept ({args.masking_mode} mask); step 1 took {t1 - t0:.2f} s", flush=True)
    say(tree, flush=True)
    say("* Step 2: AA", args.samples_per_cell, flush=True)
    step2(args, tree, model, state, comm, seed=args.seed)
    tree.relu_sigma_()
    tree.shrink_to_fit()
    torch.cuda.synchronize(); comm.barrier(); t2 = time.time()
    say(f"  step 2 took {t2 - t1:.2f} s", flush=True)
    say(tree, flush=True)
    if comm.rank == 0:
        base = os.path.dirname(args.output)
        if base:
            os.makedirs(base, exist_ok=True)
        print("* Saving", args.output, flush=True)
        tree.save(args.output, compress=False)
    if args.eval:
        test = datasets.get_dataset("test", args, device)
        say("* Evaluation (before fine tune)", flush=True)
        psnr, _ = eval_octree(tree, test, args, comm)
        say("Average PSNR", psnr, flush=True)
    comm.shutdown()
    return tree


if __name__ == "__main__":
    main(sys.argv[1:])
