"""Octree fine-tuning on the MI355X path (reference: octree/optimization.py): optimises the tree data against
the training images through the octree renderer.

Per image (reference :214-230): render -> clamp -> MSE -> backward -> optimizer step.  Here: HIP render
forward, HIP clamped-MSE gradient, HIP render backward (atomic scatter into the tree gradient), HIP SGD/Adam.
With N GPUs each image's gradient is computed on one rank in turn and summed with one RCCL all-reduce per
group of N images; N = 1 reproduces the reference's per-image steps.  `--dp_grad_reduce mean` (default) averages the
group's gradients: an N-image mini-batch at the reference's learning rate (N times fewer, equally long steps per
epoch -- safe at the reference's lr of ~1e7, which is tuned for single-image steps).  `sum` applies the SUM at the
unchanged lr -- to first order the reference's N consecutive per-image steps, i.e. an N times larger step from one
point; it can overshoot or trip the "Stop since overfitting" break and is opt-in (a warning is printed).  Adam is
invariant to the choice.

    python -m plenoctree_amd.octree.optimization --input tree.npz --output tree_opt.npz --config blender --data_dir ...
"""
import os
import sys

import numpy as np
import torch

from .. import dist, ops
from .. import octree_ops as oops
from ..nerf_sh.nerf import datasets, utils
from .svox import N3Tree, VolumeRenderer


def define_flags():
    """octree/optimization.py:60-139 + octree/nerf/utils.py:211-219."""
    p = utils.define_flags()
    a = p.add_argument
    a("--input", type=str, default="./tree.npz")
    a("--output", type=str, default="./tree_opt.npz")
    a("--render_interval", type=int, default=0)               # :71-74: every n-th validation image is also written out
    a("--val_interval", type=int, default=2)
    a("--num_epochs", type=int, default=80)
    a("--sgd", type=utils._bool, default=True)
    a("--lr", type=float, default=1e7)
    a("--sgd_momentum", type=float, default=0.0)
    a("--sgd_nesterov", type=utils._bool, default=False)
    a("--split_train", type=utils._bool, default=None)
    a("--split_holdout_prop", type=float, default=0.2)
    a("--write_vid", type=str, default=None)                  # octree/optimization.py:99-103: defined there, read nowhere
    a("--nosave", action="store_true")
    a("--continue_on_decrease", action="store_true")
    a("--renderer_step_size", type=float, default=1e-4)
    a("--no_early_stop", action="store_true")
    a("--dp_grad_reduce", type=str, default="mean", choices=["sum", "mean"])     # multi-GPU only (see module doc)
    return p


class TreeOptimizer:
    """SGD (torch.optim.SGD semantics) or Adam (torch.optim.Adam, eps 1e-8) on tree.data; octree/optimization.py:176-187."""

    def __init__(self, tree, args):
        self.tree, self.args = tree, args
        self.step_count = 0
        self.grad = torch.zeros_like(tree.data)
        self.buf = torch.zeros_like(tree.data) if (args.sgd and args.sgd_momentum != 0.0) else None
        if not args.sgd:
            self.m, self.v = torch.zeros_like(tree.data), torch.zeros_like(tree.data)

    def zero_grad(self):
        self.grad.zero_()

    def step(self, grad_scale=1.0):
        a = self.args
        if a.sgd:
            oops.sgd_step(self.tree.data.data, self.grad, a.lr * grad_scale, a.sgd_momentum, a.sgd_nesterov, self.buf,
                          first_step=self.step_count == 0)
        else:
            ops.adam_step(self.tree.data.data.view(-1), self.m.view(-1), self.v.view(-1), self.grad.view(-1), a.lr,
                          self.step_count, grad_scale=grad_scale)
        self.step_count += 1


@torch.no_grad()
def train_image(renderer, opt, c2w, gt, H, W, focal):
    """One image: accumulates d mse / d data into opt.grad and returns the device scalar sum of squares."""
    im = renderer.render_persp(c2w, width=W, height=H, fx=focal, fast=False)
    sse, g = oops.image_mse(im, gt, want_grad=True)
    oops.octree_render_persp_bwd(renderer.tree.view(), c2w, W, H, focal, renderer._opts(False), g, opt.grad, out_rgb=im)
    return sse


@torch.no_grad()
def run_validation(renderer, c2ws, images, H, W, focal, comm, vis=None):
    """run_test_step (:189-208).  vis = (directory, step index, interval): every interval-th image is written as
    `<dir>/<step>_<j>.png`, ground truth and render side by side (:202-205)."""
    acc = torch.zeros(2, dtype=torch.float64, device=renderer.tree.device)
    for j in range(comm.rank, len(c2ws), comm.world):
        im = renderer.render_persp(c2ws[j], width=W, height=H, fx=focal, fast=False)
        if vis is not None and vis[2] > 0 and j % vis[2] == 0:
            from PIL import Image
            pair = torch.cat((images[j], im.clamp(0.0, 1.0)), dim=1)
            Image.fromarray((pair * 255).to(torch.uint8).cpu().numpy()).save(os.path.join(vis[0], f"{vis[1]:04}_{j:04}.png"))
        sse, _ = oops.image_mse(im, images[j], want_grad=False)
        acc[0] += -10.0 * torch.log10(sse.double().reshape(()) / im.numel())    # on the device: no host sync per image
        acc[1] += 1
    comm.all_reduce_sum(acc)
    return float(acc[0] / acc[1])


@torch.no_grad()
def fit(args, tree, train, val, H, W, focal, comm, say=print):
    """The epoch loop of octree/optimization.py:189-243.  train/val = (c2w [n,4,4], list of [H,W,3] images).
    Rank r takes image j0 + r of every group of `world` images; the group's gradients are summed with one
    all-reduce and applied as one step (on their sum or mean, --dp_grad_reduce).  Returns (history, best tree on the CPU or None)."""
    (train_c2w, train_gt), (test_c2w, test_gt) = train, val
    renderer = VolumeRenderer(tree, step_size=args.renderer_step_size)
    opt = TreeOptimizer(tree, args)
    say("Using SGD, lr" if args.sgd else "Using Adam, lr", args.lr, flush=True)
    if comm.world > 1 and args.sgd and getattr(args, "dp_grad_reduce", "mean") == "sum":
        say(f"warning: --dp_grad_reduce sum applies {comm.world} images' summed gradient at lr {args.lr} "
            "(the reference's lr is tuned for single-image steps)", flush=True)
    vis_dir = None
    if getattr(args, "render_interval", 0) > 0:
        vis_dir = os.path.splitext(args.input)[0] + "_render"                    # :163-164
        os.makedirs(vis_dir, exist_ok=True)
    vis = lambda step: None if vis_dir is None else (vis_dir, step, args.render_interval)
    best = run_validation(renderer, test_c2w, test_gt, H, W, focal, comm, vis(0))
    say("** initial val psnr ", best, flush=True)
    history, best_tree = [(0, None, best)], None
    n_train = len(train_gt)
    for epoch in range(args.num_epochs):
        tpsnr = torch.zeros(1, dtype=torch.float64, device=tree.device)
        for j0 in range(0, n_train, comm.world):
            j = j0 + comm.rank
            opt.zero_grad()
            if j < n_train:
                sse = train_image(renderer, opt, train_c2w[j], train_gt[j], H, W, focal)
                tpsnr += -10.0 * torch.log10(sse.double().reshape(()) / (H * W * 3))   # device-side, read once per epoch
            n_imgs = min(comm.world, n_train - j0)
            comm.all_reduce_sum(opt.grad)
            opt.step(grad_scale=1.0 if getattr(args, "dp_grad_reduce", "mean") == "sum" else 1.0 / n_imgs)
        comm.all_reduce_sum(tpsnr)
        train_psnr = float(tpsnr) / n_train
        say("epoch", epoch, "** train_psnr", train_psnr, flush=True)
        if epoch % args.val_interval == args.val_interval - 1 or epoch == args.num_epochs - 1:
            val_psnr = run_validation(renderer, test_c2w, test_gt, H, W, focal, comm, vis(epoch + 1))
            say("** val psnr ", val_psnr, "best", best, flush=True)
            history.append((epoch + 1, train_psnr, val_psnr))
            if val_psnr > best:
                best, best_tree = val_psnr, tree.clone(device="cpu")
            elif not args.continue_on_decrease:
                say("Stop since overfitting", flush=True)
                break
    return history, best_tree


def main(argv=None):
    args = define_flags().parse_args(argv)
    utils.update_flags(args)
    if not torch.cuda.is_available():
        raise SystemExit("octree.optimization needs a ROCm GPU; the HIP path has no CPU fallback")
    comm = dist.init_from_env()
    torch.cuda.set_device(comm.local_rank)
    device = torch.device("cuda", comm.local_rank)
    torch.manual_seed(20200823)
    say = print if comm.rank == 0 else (lambda *a, **k: None)

    def get_data(stage):
        ds = datasets.get_dataset(stage, args, device)
        c2w = torch.from_numpy(np.ascontiguousarray(ds.camtoworlds)).float().to(device)
        gt = [ds.get_image(i)["pixels"].contiguous() for i in range(ds.size)]
        return ds, c2w, gt

    ds, train_c2w, train_gt = get_data("train")
    H, W, focal = ds.h, ds.w, ds.focal
    if args.split_train:
        test_sz = int(train_c2w.shape[0] * args.split_holdout_prop)
        say("Splitting train to train/val manually, holdout", test_sz, flush=True)
        perm = torch.randperm(train_c2w.shape[0]).tolist()          # same on every rank (seeded above)
        test_c2w, test_gt = train_c2w[perm[:test_sz]], [train_gt[i] for i in perm[:test_sz]]
        train_c2w, train_gt = train_c2w[perm[test_sz:]], [train_gt[i] for i in perm[test_sz:]]
    else:
        _, test_c2w, test_gt = get_data("test" if args.dataset == "synthetic" else "val")
    say("N3Tree load", flush=True)
    tree = N3Tree.load(args.input, map_location=device)
    history, best_tree = fit(args, tree, (train_c2w, train_gt), (test_c2w, test_gt), H, W, focal, comm, say)
    if not args.nosave and comm.rank == 0:
        if best_tree is not None:
            print("Saving best model to", args.output, flush=True)
            best_tree.save(args.output, compress=False)
        else:
            print("Did not improve upon initial model", flush=True)
    comm.shutdown()
    return history


if __name__ == "__main__":
    main(sys.argv[1:])
