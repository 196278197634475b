#!/usr/bin/env python
"""Headline benchmark: NeRF-SH training rays/sec (800x800 images, 64 coarse + 128 fine samples)
on N MI355X, BASELINE.json configs[1] (chair-shaped SH16 workload, 4096 rays per GPU per step).

    python bench.py                      # 1 GPU, 100 timed steps
    python bench.py --gpus 8             # spawns 8 ranks itself (torch.distributed.run, RCCL)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one full train_step (nerf_sh/train.py:51-121) on a synthetic batch: batch sampling,
weight re-pack, coarse+fine forward, losses (incl. the 10k-point sparsity branch), backward,
ONE RCCL all-reduce of the 4.05 MB gradient arena + stats (N > 1), Adam.  Rank 0 prints ONE JSON
line; besides the headline it carries three more records of the other BASELINE configs:
`tt_sh25` (configs[3] shape), `render_fwd` (the eval path) and `grid512` (configs[4]).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_FWD_PER_ROW = {3: 1007104, 4: 1020928}          # GEMM MAC x2 per sample
FLOP_TRAIN_PER_RAY = {3: 756.9e6, 4: 767.6e6}
FLOP_RENDER_PER_RAY = {3: 257.8e6, 4: 261.4e6}
FLOP_SIGMA_PER_POINT = 982528                        # trunk + sigma head only (131.9 TFLOP at 512^3)
PEAK_F32_MFMA_TFLOPS = 157.3                         # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32


def parse(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=100)
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--batch", type=int, default=4096, help="rays per GPU per step (weak) / global (strong)")
    p.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    p.add_argument("--preset", choices=["blender", "tt"], default="blender")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-extras", action="store_true", help="skip the tt_sh25 / render_fwd / grid512 records")
    p.add_argument("--force-dist", action="store_true",
                   help="initialise RCCL and issue the per-step collectives even with one rank (exercises the "
                        "multi-GPU code path on a 1-GPU box)")
    p.add_argument("--no-kernel-events", action="store_true",
                   help="A/B only: no HIP events around the dominant kernels in the timed region (no roofline leg)")
    p.add_argument("--cpu-rays", type=int, default=512, help="rays in the bounded CPU-baseline sample")
    p.add_argument("--cpu-steps", type=int, default=8)
    return p.parse_args(argv)


def launch_command(n_gpus, argv, port):
    """The torch.distributed.run command bench.py re-executes itself through for --gpus N > 1."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def self_launch(a):
    """`python bench.py --gpus N` without a launcher: one rank per GPU over RCCL."""
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < a.gpus:
        raise SystemExit(f"bench.py --gpus {a.gpus}: only {have} ROCm device(s) visible")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC (RCCL between processes)
    return subprocess.call(launch_command(a.gpus, sys.argv[1:], port), env=env)


def flags_for(preset, batch):
    from plenoctree_amd.nerf_sh.nerf import utils
    args = utils.define_flags().parse_args([])
    args.config = preset
    utils.update_flags(args)
    args.dataset = "synthetic"
    args.batch_size = batch
    args.factor = 0                       # 800 x 800
    args.train_dir = "/tmp/pxo_bench"
    return args


def cpu_baseline(args_ns, n_rays, n_steps, device):
    """The oracle (CPU restatement of the reference graph, not JAX) timed on this host's cores
    on a bounded sample of the same workload: `n_rays` rays x (64+128) samples + 10k sparsity
    points, forward + backward + Adam, float32, torch CPU threads = all cores."""
    from oracle import nerf_oracle as O
    from plenoctree_amd.nerf_sh.nerf import datasets
    cfg = O.Cfg(sh_deg=args_ns.sh_deg, near=args_ns.near, far=args_ns.far,
                sparsity_length=args_ns.sparsity_length, sparsity_radius=args_ns.sparsity_radius)
    gen = torch.Generator().manual_seed(0)
    # the same feeder as the GPU legs (batches are drawn on the device and copied to the host before the clock starts)
    ds_dev = datasets.Synthetic("train", args_ns, device, batch_size=n_rays)

    class _HostBatches:
        def __next__(self):
            b = next(ds_dev)
            return {"rays": type(b["rays"])(*[t.cpu() for t in b["rays"]]), "pixels": b["pixels"].cpu()}
    ds = _HostBatches()

    def step_once(flat, m, v, step, batch):
        rays = O.Rays(*batch["rays"])
        n = rays.origins.shape[0]
        t_rand = torch.rand(n, 64, generator=gen); u = torch.rand(n, 128, generator=gen)
        sp = (torch.rand(cfg.sparsity_npoints, 3, generator=gen) * 2 - 1) * cfg.sparsity_radius
        t0 = time.perf_counter()
        out = O.train_step(flat, m, v, step, rays, batch["pixels"], cfg, t_rand, u, sp, 5e-4)
        return out[:3], time.perf_counter() - t0

    flat = O.flatten_params(O.init_params(cfg))
    m = torch.zeros_like(flat); v = torch.zeros_like(flat)
    # give the CPU its best thread count: torch's intra-op pool oversubscribes badly on many-core
    # hosts, so a 32-ray probe picks among a few candidates (each probe ~1 s)
    ncpu = os.cpu_count() or 1
    small = {k: (type(val)(*[t[:32] for t in val]) if k == "rays" else val[:32]) for k, val in next(ds).items()}
    best, cores = None, 1
    for c in sorted({min(ncpu, x) for x in (8, 16, 32, 64, 128)}):
        torch.set_num_threads(c)
        step_once(flat, m, v, 0, small)
        _, dt = step_once(flat, m, v, 0, small)
        if best is None or dt < best:
            best, cores = dt, c
    torch.set_num_threads(cores)
    times = []
    for step in range(n_steps + 1):
        (flat, m, v), dt = step_once(flat, m, v, step, next(ds))
        if step > 0:                       # first step warms the allocator / thread pool
            times.append(dt)
        if sum(times) + dt > 45.0 and times:   # keep the default run within a few minutes
            break
    rps = n_rays * len(times) / sum(times)
    return {"value": rps, "unit": "rays/s", "cores": cores, "kind": "port",
            "sample": f"{len(times)} train steps of {n_rays} rays x (64+128) samples + {cfg.sparsity_npoints} "
                      f"sparsity points, oracle/nerf_oracle.py (torch-CPU f32 restatement, not JAX), "
                      f"{sum(times):.1f} s"}


def hbm_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed PMC passes of this same command
    (profiles/hbm_traffic.json; rocprofv3 cannot run inside the timed process).  None if absent."""
    try:
        with open(os.path.join(ROOT, "profiles", "hbm_traffic.json")) as f:
            k = json.load(f)["kernels"][kernel]
        return k["hbm_read_bytes_per_launch"] + k["hbm_write_bytes_per_launch"]
    except Exception:
        return None


class Job:
    """Rank context shared by the legs of the benchmark."""

    def __init__(self, a):
        self.a = a
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(self.local_rank)
        self.device = torch.device("cuda", self.local_rank)
        self.dist = None
        self.ranks_seen = 1
        if self.world > 1 or getattr(a, "force_dist", False):
            import torch.distributed as dist_mod
            self.dist = dist_mod
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            self.dist.init_process_group("nccl", rank=self.rank, world_size=self.world,
                                         device_id=self.device)                    # RCCL over xGMI
            one = torch.ones(1, device=self.device)
            self.dist.all_reduce(one)
            self.ranks_seen = int(one.item())

    def all_reduce_sum(self, t):
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)

    def sync(self):
        if self.dist:
            self.dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(self, seconds):
        if not self.dist:
            return seconds
        t = torch.tensor([seconds], dtype=torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def comm(self):
        from plenoctree_amd import dist as pdist
        return pdist.Comm(self.world, self.rank, self.local_rank, "nccl" if self.dist else None)


def read_kernels(ops, deg):
    """Live HIP-event timings of the dominant kernels since profile_enable (this rank)."""
    names = ["mlp_fwd_kernel", "mlp_bwd_data_kernel", "wgrad_kernel<256,256>", "wgrad_kernel<other>"]
    # algorithmic FLOP per row: fwd 1.007 MFLOP; bwd(data) = 7 x 2*256*256 + 2*256*(3K+1);
    # wgrad 256x256 = 2*256*256 per launch-row
    C1 = 3 * (deg + 1) ** 2 + 1
    flop_row = [FLOP_FWD_PER_ROW[deg], 7 * 2 * 256 * 256 + 2 * 256 * C1, 2 * 256 * 256, None]
    kernels = []
    for tag in range(4):
        n, ms, rows = ops.profile_read(tag)
        if n == 0:
            continue
        ent = {"kernel": names[tag], "launches": n, "avg_ms": ms / n, "rows_per_launch": rows / n}
        if flop_row[tag]:
            ent["tflops"] = rows * flop_row[tag] / (ms * 1e-3) / 1e12
        kernels.append(ent)
    return kernels


EVAL_STEP = 105      # render_fwd / grid512 are evaluated on the parameters after exactly this many train steps


def run_train(job, preset, steps, warmup, per_gpu=None, snapshot_step=None):
    """W untimed + K timed train steps of `preset`; returns a dict with elapsed (max over ranks), kernels, stats.
    snapshot_step: a copy of the parameters after exactly that many steps from the fixed-seed initialisation is kept
    (taken inside the run if it gets that far, by untimed extra steps otherwise), so that the records evaluated on
    "a trained-ish network" do not depend on --steps / --warmup."""
    from plenoctree_amd import ops
    from plenoctree_amd.nerf_sh.nerf import datasets, models, utils
    a = job.a
    if per_gpu is None:
        per_gpu = a.batch if a.scaling == "weak" else a.batch // job.world
    args = flags_for(preset, per_gpu)
    model, params = models.construct_nerf(args, job.device)
    state = models.TrainState(model.cfg, params)
    dataset = datasets.Synthetic("train", args, job.device, batch_size=per_gpu, seed=20201473 + job.rank)
    snap = {}

    def one_step(step):
        batch = next(dataset)
        lr = utils.learning_rate_decay(step, args.lr_init, args.lr_final, args.max_steps)
        models.train_step(model, state, batch, lr, randomized=True, seed=(step << 8) | job.rank,
                          world_size=job.world, all_reduce=job.all_reduce_sum if job.dist else None)
        if state.step == snapshot_step:
            snap["params"] = state.params.clone()          # 4 MB device copy

    for s in range(warmup):
        one_step(s)
    job.sync()
    ops.profile_enable(not getattr(a, "no_kernel_events", False))
    t0 = time.perf_counter()
    for s in range(warmup, warmup + steps):
        one_step(s)
    job.sync()
    elapsed = time.perf_counter() - t0
    ops.profile_enable(False)
    elapsed = job.max_over_ranks(elapsed)
    deg = model.cfg.sh_deg
    out = {"elapsed": elapsed, "per_gpu": per_gpu, "deg": deg, "kernels": read_kernels(ops, deg),
           "stats": dict(zip(utils.Stats._fields, state.stats.cpu().tolist())), "args": args,
           "model": model, "state": state, "dataset": dataset}
    if snapshot_step is not None:
        for s in range(state.step, snapshot_step):          # untimed: a short run did not get there
            one_step(s)
        out["eval_state"] = models.TrainState(model.cfg, snap["params"])
        out["eval_step"] = snapshot_step
    return out


def run_strong512(job, a):
    """The strong-scaling shape of BASELINE configs[2]: the reference's global batch of 4096 rays over 8 GPUs = 512
    rays per GPU per step (train.py:117-118: one pmean per step).  Measured on this job's GPUs with 512 rays each and
    the per-step collective issued through RCCL even with one rank (the gradient arena + stats, one all-reduce), so
    the number is what one GPU of an 8-GPU strong-scaling run does before the wire time of that all-reduce."""
    import torch.distributed as dist_mod
    own_group = False
    if job.dist is None:
        try:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(29533 + os.getpid() % 2000))
            dist_mod.init_process_group("nccl", rank=0, world_size=1, device_id=job.device)
            job.dist, own_group = dist_mod, True
        except Exception as e:                       # the record then says so instead of failing the bench
            own_group = None
            err = repr(e)[:200]
    k = max(40, a.steps)
    t = run_train(job, a.preset, k, 5, per_gpu=512)
    rec = {"value": 512 * job.world * k / t["elapsed"], "unit": "rays/s", "rays_per_gpu": 512, "steps": k,
           "ms_per_step": 1e3 * t["elapsed"] / k, "collectives_per_step": 1 if job.dist else 0,
           "frac": 512 * k / t["elapsed"] * FLOP_TRAIN_PER_RAY[t["deg"]] / (PEAK_F32_MFMA_TFLOPS * 1e12),
           "kernels": [{"kernel": e["kernel"], "avg_ms": e["avg_ms"], "tflops": e.get("tflops")} for e in t["kernels"]],
           "note": "per-GPU work of the 8-GPU strong-scaling run of the reference's 4096-ray batch; 10k sparsity points "
                   "per GPU per step (train.py:78-80) are not counted as rays"}
    if own_group:
        job.sync()
        dist_mod.destroy_process_group()
        job.dist = None
    elif own_group is None:
        rec["rccl_init_error"] = err
    return rec


def split_precision_twin(tr):
    """Model/state pair with the same parameters and PxoCfg.mlp_precision = bf16x3 (opt-in inference path)."""
    from plenoctree_amd.nerf_sh.nerf import models
    cfg = type(tr["model"].cfg).from_buffer_copy(tr["model"].cfg)
    cfg.mlp_precision = 1
    twin = dict(tr)
    twin["model"] = models.NerfModel(cfg)
    twin["eval_state"] = models.TrainState(cfg, tr["eval_state"].params.clone())
    return twin


def run_render(job, tr, iters=20):
    """The eval path (nerf_sh/eval.py -> utils.render_image): pxo_render_fwd on `batch` rays per GPU per call,
    deterministic sampling (eval.py:57)."""
    from plenoctree_amd import ops
    model, state = tr["model"], tr["eval_state"]
    batch = next(tr["dataset"])
    rays = batch["rays"]
    for _ in range(2):
        model.apply(state, rays, False)
    job.sync()
    ops.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(iters):
        model.apply(state, rays, False)
    job.sync()
    elapsed = job.max_over_ranks(time.perf_counter() - t0)
    ops.profile_enable(False)
    read_kernels(ops, tr["deg"])          # drain the event records
    rps = tr["per_gpu"] * job.world * iters / elapsed
    return {"value": rps, "unit": "rays/s", "calls": iters, "rays_per_call_per_gpu": tr["per_gpu"],
            "ms_per_call": 1e3 * elapsed / iters, "params_after_steps": tr["eval_step"],
            "frac": rps / job.world * FLOP_RENDER_PER_RAY[tr["deg"]] / (PEAK_F32_MFMA_TFLOPS * 1e12)}


def run_grid512(job, tr, stages_after_grid=True):
    """BASELINE configs[4]: step 1 of octree.extraction at init_grid_depth 8 (octree/extraction.py:288-352):
    sigma of MLP_1 on the 512^3 grid (x-slabs sharded over the ranks + all-gather), the weight mask over the 100
    training views (cameras sharded + max-all-reduce) and the tree build."""
    from plenoctree_amd import octree_ops as oops
    from plenoctree_amd.octree import extraction
    from plenoctree_amd.octree.svox import N3Tree
    model, state, dataset = tr["model"], tr["eval_state"], tr["dataset"]
    comm = job.comm()
    reso, center, radius = 512, [0.0, 0.0, 0.0], [1.5, 1.5, 1.5]
    if model.cfg.mlp_precision == 0:
        state.repack(need_bwd=False)
    extraction.grid_sigma(model, state, 64, center, radius, comm)       # warm-up (small grid)
    job.sync()
    t0 = time.perf_counter()
    sig = extraction.grid_sigma(model, state, reso, center, radius, comm)
    job.sync()
    t_grid = job.max_over_ranks(time.perf_counter() - t0)
    if stages_after_grid is False:
        del sig
        tflops = reso ** 3 * FLOP_SIGMA_PER_POINT / t_grid / 1e12
        return {"points": reso ** 3, "grid_ms": 1e3 * t_grid, "equivalent_f32_tflops": tflops}
    tree = N3Tree(N=2, data_dim=1 + 3 * (tr["deg"] + 1) ** 2, init_refine=0, depth_limit=8, radius=radius,
                  center=center, data_format=f"SH{(tr['deg'] + 1) ** 2}", map_location=job.device)
    t0 = time.perf_counter()
    weights = extraction.calculate_grid_weights(dataset, sig, reso, tree.invradius, tree.offset, 1e-4, comm)
    job.sync()
    t_weight = job.max_over_ranks(time.perf_counter() - t0)
    # an untrained network has no surfaces: threshold sigma at its 97th percentile, i.e. a mask of ~4 M voxels as a
    # trained scene leaves (the build time depends on how many voxels are set, not on which)
    thr = float(torch.quantile(sig[::4099].float(), 0.97))
    job.sync()
    t0 = time.perf_counter()
    mask = oops.threshold_mask(sig, thr)
    tree.refine_from_mask(mask)
    job.sync()
    t_tree = job.max_over_ranks(time.perf_counter() - t0)
    n_pts = reso ** 3
    tflops = n_pts * FLOP_SIGMA_PER_POINT / t_grid / 1e12
    weight_voxels, mask_voxels = int((weights >= 1e-3).sum()), int(mask.sum())     # weight_thresh default, extraction.py:126-132
    del weights, mask, sig
    return {"points": n_pts, "params_after_steps": tr["eval_step"], "grid_ms": 1e3 * t_grid, "tflops": tflops,
            "frac": tflops / job.world / PEAK_F32_MFMA_TFLOPS, "weight_mask_ms": 1e3 * t_weight,
            "weight_mask_views": int(dataset.size), "weight_mask_voxels": weight_voxels,
            "tree_build_ms": 1e3 * t_tree, "tree_mask_voxels": mask_voxels,
            "tree_nodes": int(tree.n_internal), "sharding": f"x-slabs over {job.world} GPU(s) + all-gather"}


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(a))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus != world:
        raise SystemExit(f"bench.py --gpus {a.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU; the HIP path has no CPU fallback")
    job = Job(a)
    rank = job.rank

    from plenoctree_amd import build
    if rank == 0:
        build.build(verbose=False)
    if job.dist:
        job.dist.barrier()

    tr = run_train(job, a.preset, a.steps, a.warmup, snapshot_step=None if a.no_extras else EVAL_STEP)
    extras = {}
    if not a.no_extras:
        extras["strong512"] = run_strong512(job, a)
        extras["render_fwd"] = run_render(job, tr)
        extras["grid512"] = run_grid512(job, tr)
        # opt-in inference precision (NOT the headline, NOT used in training): products as 3 bf16 MFMAs, f32 accumulate
        twin = split_precision_twin(tr)
        r3, g3 = run_render(job, twin), run_grid512(job, twin, stages_after_grid=False)
        extras["opt_in_bf16x3_inference"] = {
            "note": "PxoCfg.mlp_precision = bf16x3: forward-only, |dPSNR| vs the f64 oracle <= 1e-4 dB (tests/test_gpu_x3.py); "
                    "training and the headline stay float32",
            "render_fwd_rays_per_s": r3["value"], "render_fwd_ms_per_call": r3["ms_per_call"],
            "grid512_ms": g3["grid_ms"], "grid512_equivalent_f32_tflops": g3["equivalent_f32_tflops"]}
        twin = None
        other = "tt" if a.preset == "blender" else "blender"
        tr["state"] = tr["eval_state"] = tr["dataset"] = None   # release the headline workspace before the second preset
        torch.cuda.empty_cache()
        k2 = max(10, a.steps // 4)
        t2 = run_train(job, other, k2, 3)
        v2 = t2["per_gpu"] * world * k2 / t2["elapsed"]
        extras["tt_sh25" if other == "tt" else "blender_sh16"] = {
            "value": v2, "unit": "rays/s", "steps": k2, "ms_per_step": 1e3 * t2["elapsed"] / k2,
            "frac": v2 / world * FLOP_TRAIN_PER_RAY[t2["deg"]] / (PEAK_F32_MFMA_TFLOPS * 1e12),
            "mlp_fwd_tflops": t2["kernels"][0]["tflops"] if t2["kernels"] else None,
            "workload": "nerf_sh/config/tt.yaml: SH25, near 0, far 4, sparsity_length 0.2, sparsity_radius 5"
                        if other == "tt" else "nerf_sh/config/blender.yaml"}
        t2 = None

    if rank == 0:
        per_gpu, deg, kernels = tr["per_gpu"], tr["deg"], tr["kernels"]
        elapsed = tr["elapsed"]
        value = per_gpu * world * a.steps / elapsed
        dom = kernels[0] if kernels else None            # mlp_fwd_kernel: largest single launch of the step
        roofline = None
        if dom:
            roofline = {"bound": "mfma", "kernel": dom["kernel"], "achieved": dom["tflops"],
                        "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": dom["tflops"] / PEAK_F32_MFMA_TFLOPS,
                        "avg_launch_ms": dom["avg_ms"], "launches": dom["launches"],
                        "flop_per_launch": dom["rows_per_launch"] * FLOP_FWD_PER_ROW[deg],
                        "traffic": hbm_traffic("mlp_fwd_kernel")}
        out = {
            "metric": "training rays/sec (800x800, 64+128 samples)",
            "value": value, "unit": "rays/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * elapsed / a.steps, "higher_is_better": True, "scaling": a.scaling,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"NeRF-SH {'SH16 blender' if a.preset == 'blender' else 'SH25 tt'} preset, "
                                   f"{per_gpu} rays/GPU/step x (64 coarse + 128 fine) samples, "
                                   "800x800 synthetic views, sparsity 10k pts, Adam",
                       "rays_per_gpu": per_gpu, "global_batch": per_gpu * world, "sh_deg": deg,
                       "parallelism": f"dp{world}"},
            "nccl_ranks_seen": job.ranks_seen, "collectives_per_step": 1 if job.dist else 0,
            "step_mfma_frac": value / world * FLOP_TRAIN_PER_RAY[deg] / (PEAK_F32_MFMA_TFLOPS * 1e12),
            "final_stats": tr["stats"],
            "roofline": roofline, "kernels": kernels,
        }
        out.update(extras)
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(tr["args"], a.cpu_rays, a.cpu_steps, job.device)
        print(json.dumps(out), flush=True)
        # RCCL keeps its version banner in the C stdio buffer until the process exits: whatever libraries still flush to
        # fd 1 after this point goes to stderr, so that the JSON line above stays the only line on stdout
        sys.stdout.flush()
        os.dup2(2, 1)
    if job.dist:
        job.dist.barrier()
        job.dist.destroy_process_group()


if __name__ == "__main__":
    main()
