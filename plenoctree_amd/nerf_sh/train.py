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
    if args.mlp_precision == "bf16x3":
        raise ValueError("--mlp_precision bf16x3 is an inference option (eval / gen_video / extraction); training runs in float32 "
                         "or in its float32-accurate emulation bf16x6")
    h0 = comm.rank == 0
    render_dir = os.path.join(args.train_dir, "render")
    timings_file = None
    if h0:
        os.makedirs(render_dir, exist_ok=True)                        # train.py:133-135
        timings_file = open(os.path.join(args.train_dir, "timings.txt"), "a")   # train.py:138-143

    def write_ts_now(step):
        if timings_file is not None:
            from datetime import datetime
            timings_file.write(f"{step} {datetime.now().isoformat()}\n")
            timings_file.flush()
    write_ts_now(0)

    per_rank = args.batch_size // comm.world
    args.per_host_image = dist.per_host_image(args.per_host_image, comm.world)
    if args.per_host_image and not args.image_batching:
        # the reference on ONE host with N local devices: one image per step, its batch_size pixels drawn once and sharded
        # (datasets.py:159-166, utils.py:518-522); every rank carries host 0's seed and takes its contiguous piece of the draw
        dataset = datasets.get_dataset("train", args, device, batch_size=per_rank, seed=20201473,
                                       shard=(comm.rank, comm.world))
    else:
        # every rank its own host (np.random.seed(20201473 + jax.host_id()), train.py:128): its own image, its own pixels
        dataset = datasets.get_dataset("train", args, device, batch_size=per_rank, seed=20201473 + comm.rank)
    test_dataset = datasets.get_dataset("test", args, device)
    model, state = models.get_model_state(args, device, restore=True)
    init_step = state.step + 1                                       # train.py:176
    if h0:
        print(f"* {2 * state.n_mlp} parameters, resuming at step {init_step}, {comm.world} GPU(s), "
              f"{per_rank} rays/GPU", flush=True)

    reducer = dist.GradReducer(comm, device)
    t_loop_start = time.time()
    stats_trace = []
    # the reference averages loss / psnr over EVERY step since the last print (stats_trace, train.py:199-200,216-218); here the
    # sum lives on the device (one 6-float add per step) so that the host never waits for a step it does not print
    stats_acc = torch.zeros(6, dtype=torch.float64, device=device)
    acc_steps = 0
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
        stats_acc.add_(state.stats)
        acc_steps += 1
        if step % args.print_every == 0:                            # train.py:208-236
            torch.cuda.synchronize()
            s = utils.Stats(*state.stats.cpu().tolist())
            avg = utils.Stats(*(stats_acc / max(acc_steps, 1)).cpu().tolist())
            stats_acc.zero_()
            acc_steps = 0
            steps_per_sec = args.print_every / (time.time() - t_loop_start)
            reset_timer = True
            rays_per_sec = args.batch_size * steps_per_sec          # train.py:224
            if h0:
                precision = int(np.ceil(np.log10(args.max_steps))) + 1
                print(("{:" + "{:d}".format(precision) + "d}").format(step) + f"/{args.max_steps:d}: "
                      + f"i_loss={s.loss:0.4f}, avg_loss={avg.loss:0.4f}, weight_l2={s.weight_l2:0.2e}, lr={lr:0.2e}, "
                      + f"{rays_per_sec:0.0f} rays/sec", flush=True)
                stats_trace.append((step, s.loss, s.psnr, rays_per_sec, avg.loss, avg.psnr))
        if step % args.save_every == 0 and h0:                      # train.py:237-242
            checkpoints.save_checkpoint(args.train_dir, state, step, keep=200)
        if args.render_every > 0 and step % args.render_every == 0:  # train.py:245-296 (PSNR only)
            ex = next(test_dataset)
            t0 = time.time()
            rgb, disp, acc = utils.render_image(
                lambda r: model.apply(state, r, args.randomized, seed=step), ex["rays"], chunk=args.chunk,
                world_size=comm.world, rank=comm.rank, gather=comm.all_gather_cat)
            torch.cuda.synchronize()
            if h0:                                                   # train.py:262-296
                write_ts_now(step)
                psnr = utils.compute_psnr(((rgb - ex["pixels"]) ** 2).mean().item())
                ssim = float(utils.compute_ssim(rgb.clamp(0.0, 1.0), ex["pixels"], max_val=1.0))
                n = ex["pixels"].shape[0] * ex["pixels"].shape[1]
                print(f"Eval {step}: {time.time() - t0:0.3f}s., {n / (time.time() - t0):0.0f} rays/sec, PSNR {psnr:.4f}, "
                      f"SSIM {ssim:.4f}", flush=True)
                # [target | prediction | disparity | accumulation] side by side, render/<step>.png (train.py:283-290)
                vis = torch.cat([ex["pixels"], rgb, disp.expand(-1, -1, 3), acc.expand(-1, -1, 3)], dim=1)
                out_path = os.path.join(render_dir, "{:010}.png".format(step))
                utils.save_img(vis, out_path)
                print(" Rendering saved to ", out_path, flush=True)
    if args.max_steps % args.save_every != 0 and h0:                # train.py:306-310
        checkpoints.save_checkpoint(args.train_dir, state, int(args.max_steps), keep=200)
    if timings_file is not None:
        timings_file.close()
    comm.barrier()
    comm.shutdown()
    return stats_trace


if __name__ == "__main__":
    main(sys.argv[1:])
