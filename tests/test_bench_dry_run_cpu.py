"""bench.py's own multi-rank control flow, executed on two gloo CPU ranks before any 8-GPU node sees it.

Each rank installs the oracle-backed stand-ins of tests/_bench_standins.py and calls bench.main() with
`--backend gloo` and small sizes, with RANK / WORLD_SIZE / MASTER_* set the way the driver's
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` sets them.  Checked: the rendezvous, ONE JSON line
from rank 0 and none from rank 1, two gradient collectives per step through dist.GradReducer, the `strong512` record
inside the EXISTING process group, the sharded records (grid all-gather, camera-sharded weight mask, sharded eval render
of `converge`), and that `cpu_baseline` stays in the line with world > 1."""
import contextlib
import io
import json
import os
import socket
import sys
import tempfile

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

ARGS = ["--backend", "gloo", "--batch", "8", "--steps", "2", "--warmup", "1", "--strong-rays", "4", "--grid-reso", "8",
        "--eval-step", "2", "--converge-steps", "2", "--converge-views", "1", "--image-factor", "100",
        "--sparsity-npoints", "16", "--cpu-rays", "8", "--cpu-steps", "1", "--cpu-warmup", "1", "--no-cpu-full", "--per-host-image",
        "--extras", "converge,strong512,render_fwd,grid512,coarse64,tt_sh25"]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, outdir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(2)
    import _bench_standins
    calls = _bench_standins.install()
    import bench
    import torch.distributed as dist
    n_collectives = {"n": 0}
    real_all_reduce = dist.all_reduce

    def counting_all_reduce(t, *a, **k):
        n_collectives["n"] += 1
        return real_all_reduce(t, *a, **k)

    dist.all_reduce = counting_all_reduce
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        rc = bench.main(ARGS + ["--gpus", str(world)])
    with open(os.path.join(outdir, f"rank{rank}.json"), "w") as f:
        json.dump({"rc": rc, "stdout": buf.getvalue(), "calls": calls, "all_reduce_calls": n_collectives["n"]}, f)


@pytest.mark.timeout(900)
def test_bench_two_rank_dry_run():
    world = 2
    with tempfile.TemporaryDirectory() as outdir:
        mp.spawn(_worker, args=(world, _free_port(), outdir), nprocs=world, join=True)
        res = [json.load(open(os.path.join(outdir, f"rank{r}.json"))) for r in range(world)]
    assert [r["rc"] for r in res] == [0, 0]
    assert res[1]["stdout"].strip() == ""                          # rank 0 alone prints
    lines = [l for l in res[0]["stdout"].splitlines() if l.strip()]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["nccl_ranks_seen"] == 2 and out["config"]["parallelism"] == "dp2"
    assert out["config"]["global_batch"] == 16 and out["scaling"] == "weak" and out["steps"] == 2
    assert out["collectives_per_step"] == 2                        # MLP_0's bucket, then MLP_1's bucket + stats
    assert out["value"] > 0 and out["ms_per_step"] > 0
    s = out["strong512"]
    assert s["rays_per_gpu"] == 4 and s["collectives_per_step"] == 2 and "rccl_init_error" not in s
    assert "cpu_baseline" in out and out["cpu_baseline"]["kind"] == "port" and out["cpu_baseline"]["value"] > 0
    cb = out["cpu_baseline"]          # BASELINE.md section 3's record: both configs[0] shapes, the host, the threads used
    assert [sh["samples"] for sh in cb["shapes"]] == ["64", "64+128"] and cb["value"] == cb["shapes"][1]["rays_per_s"]
    assert cb["nproc"] >= cb["cores"] >= 1 and cb["torch_version"] and cb["shapes_dropped"]
    c = out["converge"]
    assert out["eval_psnr"] == c["eval_psnr"] and 0.0 < c["eval_psnr"] < 60.0 and c["views"] == 1
    assert c["rays_per_step"] == 16 and c["view_size"] == [8, 8]
    g = out["grid512"]
    assert g["points"] == 512 and g["weight_mask_views"] == 100 and "2 GPU(s)" in g["sharding"] and g["tree_nodes"] >= 1
    assert set(out["coarse64"]) >= {"rays8", "workload"} and out["tt_sh25"]["value"] > 0
    assert out["render_fwd"]["params_after_steps"] == 2
    # both ranks did the same work: every stand-in was reached the same number of times, cameras were sharded
    assert res[0]["calls"] == res[1]["calls"] and res[0]["calls"]["grid_weight_render"] == 1
    assert res[0]["all_reduce_calls"] == res[1]["all_reduce_calls"] > 0
