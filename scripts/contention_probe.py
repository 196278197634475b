#!/usr/bin/env python
"""What does a REAL collective kernel do to the step?  (One GPU; no N > 1 box exists for this project.)

The dense MLP kernels run one persistent 8-wave workgroup per CU at the full register / LDS budget: nothing else can be
co-resident on a CU they hold.  An 8-rank ring all-reduce of MLP_0's 2 MB gradient half is a kernel of a few dozen
workgroups that lives for 30-50 us on RCCL's high-priority stream; it starts at the `grads0_ready` point, i.e. under the fine
level.  This probe puts exactly that beside the step: pxo_occupy_cus(k workgroups x 256 threads, 50 us) launched on the
exchange stream through dist.GradReducer's `all_reduce` hook (bucket 0 only; bucket 1 is exposed at the end of the step by
construction and is left empty here), for k in {0 (stream fork/join only), 8, 16, 32, 64}, at 512 and 4096 rays per step, with
the static tile stride and with the device tile counter (PXO_TUNE_TILE_SCHED).  Variants are interleaved over several rounds; printed: median ms per step and
the slowdown against k = 0 -- to be compared with the probe's own 50 us.
"""
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    import bench
    from plenoctree_amd import dist as pdist, ops
    from plenoctree_amd.nerf_sh.nerf import datasets, models, utils
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    side = pdist.exchange_stream(dev)            # before the first kernel (profiles/r04h_late_group.txt)
    micros = float(os.environ.get("PROBE_US", "50"))
    steps = int(os.environ.get("PROBE_STEPS", "40"))
    rounds = int(os.environ.get("PROBE_ROUNDS", "5"))
    # 0: the probe's workgroups fit BESIDE a resident fused-MLP workgroup (27 KB of LDS and 144 registers per lane are free on
    # every CU); 65536: they do not, and can only start on a CU a persistent workgroup has left (a kernel boundary)
    lds_bytes = int(os.environ.get("PROBE_LDS", "0"))
    default_sched = ops.get_tuning(ops.TUNE_TILE_SCHED)
    a = bench.parse(["--no-extras"])
    out = []
    for B in (512, 4096):
        args = bench.flags_for(a, "blender", B)
        model, params = models.construct_nerf(args, dev)
        state = models.TrainState(model.cfg, params)
        ds = datasets.Synthetic("train", args, dev, batch_size=B)
        n0 = state.bucket0.numel()
        step = [0]

        def reducer_for(k):
            def fake_all_reduce(t):
                if k > 0 and t.numel() == n0:        # bucket 0, on the side stream (current inside GradReducer.reduce)
                    ops.occupy_cus(k, 256, micros, lds_bytes=lds_bytes)
            return pdist.GradReducer(pdist.Comm(1, 0, 0, None), dev, all_reduce=fake_all_reduce)

        def run(n, red):
            for _ in range(n):
                s_ = step[0]; step[0] += 1
                lr = utils.learning_rate_decay(s_, args.lr_init, args.lr_final, args.max_steps)
                models.train_step(model, state, next(ds), lr, randomized=True, seed=s_ << 8, world_size=1, reducer=red)

        ks = (0, 8, 16, 32, 64)
        reds = {k: reducer_for(k) for k in ks}
        run(5, reds[0])
        times = {(sched, k): [] for sched in (0, 1) for k in ks}
        for _ in range(rounds):                       # interleaved: box drift hits every variant alike
            for sched in (0, 1):
                ops.set_tuning(ops.TUNE_TILE_SCHED, sched)
                for k in ks:
                    run(3, reds[k])
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    run(steps, reds[k])
                    torch.cuda.synchronize()
                    times[(sched, k)].append(1e3 * (time.perf_counter() - t0) / steps)
        for sched in (0, 1):
            base = statistics.median(times[(sched, 0)])
            for k in ks:
                med = statistics.median(times[(sched, k)])
                rec = {"rays": B, "tile_sched": "counter" if sched else "static", "probe_workgroups": k, "probe_us": micros, "probe_lds_bytes": lds_bytes,
                       "median_ms_per_step": round(med, 4), "slowdown_us": round(1e3 * (med - base), 1),
                       "runs_ms": [round(x, 4) for x in times[(sched, k)]]}
                out.append(rec)
                print(json.dumps(rec), flush=True)
    ops.set_tuning(ops.TUNE_TILE_SCHED, default_sched)


if __name__ == "__main__":
    main()
