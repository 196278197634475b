#!/usr/bin/env python
"""In-process A/B of pxo_set_tuning variants: one model / dataset / workspace per batch size, variants interleaved over
several rounds (no process start-up or box-to-box spread between the things compared).
    python scripts/tune_ab.py --batches 512,4096 base: ts1:tile_sched=1 r73:wgrad_ranges=73 ...
Prints one line per (batch, variant): median and minimum ms per step."""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--batches", default="512,4096")
    p.add_argument("--steps", type=int, default=50)
    p.add_argument("--rounds", type=int, default=4)
    p.add_argument("--skip-zero-rows", action="store_true")
    p.add_argument("variants", nargs="+")
    a = p.parse_args()
    import bench
    from plenoctree_amd import ops
    from plenoctree_amd.nerf_sh.nerf import datasets, models, utils
    knobs = {"tile_sched": ops.TUNE_TILE_SCHED, "wgrad_ranges": ops.TUNE_WGRAD_RANGES,
             "wgrad_skinny_ranges": ops.TUNE_WGRAD_SKINNY_RANGES}
    variants = []
    for v in a.variants:
        name, _, spec = v.partition(":")
        variants.append((name, {k: int(x) for k, _, x in (kv.partition("=") for kv in spec.split(",") if kv)}))
    dev = torch.device("cuda", 0)
    ba = bench.parse(["--no-extras"])
    for B in [int(x) for x in a.batches.split(",")]:
        args = bench.flags_for(ba, "blender", B, skip_zero_rows=a.skip_zero_rows)
        model, params = models.construct_nerf(args, dev)
        state = models.TrainState(model.cfg, params)
        ds = datasets.Synthetic("train", args, dev, batch_size=B)
        step = [0]

        def run(n):
            for _ in range(n):
                s_ = step[0]; step[0] += 1
                lr = utils.learning_rate_decay(s_, args.lr_init, args.lr_final, args.max_steps)
                models.train_step(model, state, next(ds), lr, randomized=True, seed=s_ << 8)
        run(10)
        times = {name: [] for name, _ in variants}
        for _ in range(a.rounds):
            for name, kv in variants:
                for k in knobs:
                    ops.set_tuning(knobs[k], kv.get(k, 0))
                run(3)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                run(a.steps)
                torch.cuda.synchronize()
                times[name].append(1e3 * (time.perf_counter() - t0) / a.steps)
        for name, kv in variants:
            t = times[name]
            print(json.dumps({"rays": B, "variant": name, "tuning": kv, "median_ms": round(statistics.median(t), 4),
                              "min_ms": round(min(t), 4), "rays_per_s": round(B / statistics.median(t) * 1e3),
                              "runs_ms": [round(x, 4) for x in t]}), flush=True)
    for k in knobs.values():
        ops.set_tuning(k, 0)


if __name__ == "__main__":
    main()
