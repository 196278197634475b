"""Training entry point: `python -m plenoctree_amd.nerf_sh.train --train_dir D --config blender
--data_dir DATA` (reference: `python -m nerf_sh.train`, nerf_sh/train.py:124-310, README.md:58-67).
Multi-GPU: `python -m torch.distributed.run --nproc-per-node N -m plenoctree_amd.nerf_sh.train ...`;
batch_size is the GLOBAL batch, sharded over ranks (nerf_sh/nerf/datasets.py:80, utils.shard)."""
import os
import sys
import time

import numpy as np
import torch

from .. import dist
from .nerf import checkpoints, datasets, models, utils


def main(argv=None):
    args = utils.define_flags().parse_args(argv)
    utils.update_flags(args)
    if not torch.cuda.is_available():
        raise SystemExit("nerf_sh.train needs a ROCm GPU; the HIP path has no CPU fallback")
    comm = dist.init_from_env()
    torch.cuda.set_device(comm.local_rank)
    device = torch.device("cuda", comm.local_rank)
    utils.check_flags(args, require_batch_size_div=True, world_size=comm.world)
    if args.mlp_precision != "f32":
        raise ValueError("--mlp_precision bf16x3 is an inference option (eval / gen_video / extraction); training runs in float32")
    h0 = comm.rank == 0
    if h0:
        os.makedirs(args.train_dir, exist_ok=True)

    per_rank = args.batch_size // comm.world
    dataset = datasets.get_dataset("train", args, device, batch_size=per_rank)
    dataset.rng = np.random.RandomState(20201473 + comm.rank)      # train.py:128
    dataset.seed = 20201473 + comm.rank
    test_dataset = datasets.get_dataset("test", args, device)
    model, state = models.get_model_state(args, device, restore=True)
    init_step = state.step + 1                                       # train.py:176
    if h0:
        print(f"* {2 * state.n_mlp} parameters, resuming at step {init_step}, {comm.world} GPU(s), "
              f"{per_rank} rays/GPU", flush=True)

    reducer = dist.GradReducer(comm, device)
    t_loop_start = time.time()
    stats_trace = []
    reset_timer = True
    for step in range(init_step, args.max_steps + 1):
        if reset_timer:
            torch.cuda.synchronize()
            t_loop_start = time.time()
            reset_timer = False
        batch = next(dataset)
        lr = utils.learning_rate_decay(step, args.lr_init, args.lr_final, args.max_steps, args.lr_delay_steps,
                                       args.lr_delay_mult)
        models.train_step(model, state, batch, lr, randomized=args.randomized, seed=(step << 8) | comm.rank,
                          world_size=comm.world, reducer=reducer)
        if step % args.print_every == 0:                            # train.py:208-236
            torch.cuda.synchronize()
            s = utils.Stats(*state.stats.cpu().tolist())
            steps_per_sec = args.print_every / (time.time() - t_loop_start)
            reset_timer = True
            rays_per_sec = args.batch_size * steps_per_sec          # train.py:224
            if h0:
                precision = int(np.ceil(np.log10(args.max_steps))) + 1
                print(("{:" + "{:d}".format(precision) + "d}").format(step) + f"/{args.max_steps:d}: "
                      + f"i_loss={s.loss:0.4f}, avg_loss={s.loss:0.4f}, weight_l2={s.weight_l2:0.2e}, lr={lr:0.2e}, "
                      + f"{rays_per_sec:0.0f} rays/sec", flush=True)
                stats_trace.append((step, s.loss, s.psnr, rays_per_sec))
        if step % args.save_every == 0 and h0:                      # train.py:237-242
            checkpoints.save_checkpoint(args.train_dir, state, step, keep=200)
        if args.render_every > 0 and step % args.render_every == 0:  # train.py:245-296 (PSNR only)
            ex = next(test_dataset)
            t0 = time.time()
            rgb, disp, acc = utils.render_image(
                lambda r: model.apply(state, r, args.randomized, seed=step), ex["rays"], chunk=args.chunk,
                world_size=comm.world, rank=comm.rank, gather=comm.all_gather_cat)
            torch.cuda.synchronize()
            if h0:
                psnr = utils.compute_psnr(((rgb - ex["pixels"]) ** 2).mean().item())
                n = ex["pixels"].shape[0] * ex["pixels"].shape[1]
                print(f"Eval {step}: {time.time() - t0:0.3f}s., {n / (time.time() - t0):0.0f} rays/sec, PSNR {psnr:.4f}",
                      flush=True)
    if args.max_steps % args.save_every != 0 and h0:                # train.py:306-310
        checkpoints.save_checkpoint(args.train_dir, state, int(args.max_steps), keep=200)
    comm.barrier()
    comm.shutdown()
    return stats_trace


if __name__ == "__main__":
    main(sys.argv[1:])
