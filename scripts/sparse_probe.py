#!/usr/bin/env python
"""Per-kernel times of the train step at a TRAINED state with the dense reverse pass and with PxoCfg.skip_zero_rows:
`--train-steps` dense steps from the fixed-seed initialisation (bench.py's `converge` trajectory), then 20 steps each way
with every kernel tag bracketed by HIP events.  Prints one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    n_train = int(sys.argv[sys.argv.index("--train-steps") + 1]) if "--train-steps" in sys.argv else 2000
    a = bench.parse(["--no-cpu-baseline", "--no-extras"])
    job = bench.Job(a)
    from plenoctree_amd import ops
    tr = bench.run_train(job, a.preset, n_train, 0, events=False)
    model, state = tr["model"], tr["state"]
    out = {"after_steps": n_train, "train_psnr_last_batch": tr["stats"]["psnr"]}
    step = n_train
    for skip in (0, 1):
        model.cfg.skip_zero_rows = skip
        for s in range(step, step + 3):
            tr["one_step"](s)
        step += 3
        job.sync()
        ops.profile_enable(True)
        import time
        t0 = time.perf_counter()
        for s in range(step, step + 20):
            tr["one_step"](s)
        job.sync()
        dt = time.perf_counter() - t0
        ops.profile_enable(False)
        step += 20
        ks = bench.read_kernels(ops, tr["deg"])
        live, total = ops.train_backward_work(model.cfg, tr["per_gpu"], state._ws)
        out["skip" if skip else "dense"] = {"ms_per_step": 1e3 * dt / 20, "live_chunk_fraction": live / total,
                                            "kernels_ms_per_step": {k["kernel"]: k["avg_ms"] * k["launches"] / 20 for k in ks}}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
