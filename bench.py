#!/usr/bin/env python
"""Headline benchmark: NeRF-SH training rays/sec (800x800 images, 64 coarse + 128 fine samples)
on N MI355X, BASELINE.json configs[1] (chair-shaped SH16 workload, 4096 rays per GPU per step).

    python bench.py                      # 1 GPU, 100 timed steps
    python bench.py --gpus 8             # spawns 8 ranks itself (torch.distributed.run, RCCL)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one full train_step (nerf_sh/train.py:51-121) on a synthetic batch: batch sampling,
coarse level forward + reverse, fine level forward + reverse (incl. the 10k-point sparsity branch),
the gradient exchange (N > 1: two RCCL all-reduces, MLP_0's half of the 4.05 MB arena under the fine level,
MLP_1's half + stats at the end), Adam + weight re-pack.  Rank 0 prints ONE JSON line; besides the headline it
carries records of the other BASELINE configs and of the metric's second half:
`converge` (eval PSNR on held-out 800x800 views after a fixed training budget), `strong512` (configs[2] strong-scaling
shape), `tt_sh25` (configs[3] shape), `coarse64` (configs[0] shape: 64 coarse samples only), `render_fwd` (the eval
path), `grid512` (configs[4]), `octree` (the PlenOctree-side kernels with their HBM rooflines) and the two
opt-in split-precision records (`opt_in_bf16x3_inference`; `opt_in_bf16x6_training`: the float32-ACCURATE emulation of the two
fused MLP kernels on the bf16 matrix pipe -- its own record, `dtype` "f32 (bf16x6 emulated)"; `value`, `roofline` and every
other record stay on native float32).

`--backend gloo` is a DRY RUN of this file's multi-rank control flow on CPU ranks for tests/test_bench_dry_run_cpu.py,
which installs oracle-backed stand-ins for the HIP entry points first (the product has no CPU path: without the
stand-ins every leg fails in ops._require_gpu).  Its numbers mean nothing.
"""
import argparse
import json
import math
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
FLOP_TRAIN_PER_RAY_COARSE_ONLY = {3: 189.2e6, 4: 191.9e6}
FLOP_RENDER_PER_RAY = {3: 257.8e6, 4: 261.4e6}
FLOP_SIGMA_PER_POINT = 982528                        # trunk + sigma head only (131.9 TFLOP at 512^3)
PEAK_F32_MFMA_TFLOPS = 157.3                         # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32
ALL_EXTRAS = ("converge", "strong512", "render_fwd", "grid512", "octree", "bf16x3", "bf16x6", "coarse64", "tt_sh25")
OPT_IN_EXTRAS = ("bf16x6_converge",)      # accepted by --extras, not in the default line


def parse(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=100)
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--batch", type=int, default=4096, help="rays per GPU per step (weak) / global (strong)")
    p.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    p.add_argument("--preset", choices=["blender", "tt"], default="blender")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-extras", action="store_true", help="headline only")
    p.add_argument("--extras", default=",".join(ALL_EXTRAS), help="comma-separated subset of " + ",".join(ALL_EXTRAS + OPT_IN_EXTRAS))
    p.add_argument("--force-dist", action="store_true",
                   help="initialise RCCL and issue the per-step collectives even with one rank (exercises the "
                        "multi-GPU code path on a 1-GPU box)")
    p.add_argument("--no-kernel-events", action="store_true",
                   help="A/B only: no HIP events around the dominant kernels in the timed region (no roofline leg)")
    p.add_argument("--per-host-image", choices=["auto", "true", "false"], default="auto", nargs="?", const="true",
                   help="true: batches as the reference draws them on ONE host -- one image per step, its pixels sharded over the "
                        "ranks (datasets shard=(rank, world)); false: every rank its own image (the reference's multi-host "
                        "sampler); auto (default): true when all ranks share one host")
    p.add_argument("--tune", default="", help="A/B only: pxo_set_tuning knobs, e.g. tile_sched=1,wgrad_ranges=73,wgrad_skinny_ranges=128 "
                                              "(same results, different schedule; recorded in the line as `tuning`)")
    p.add_argument("--cpu-rays", type=int, default=1024, help="rays per step of the CPU baseline (BASELINE.md section 3: 1024)")
    p.add_argument("--cpu-steps", type=int, default=10, help="timed steps of the CPU baseline (after --cpu-warmup)")
    p.add_argument("--cpu-warmup", type=int, default=3)
    p.add_argument("--no-cpu-full", action="store_true",
                   help="leave the B = 4096 shape (configs[1], ~3 minutes of CPU) out of the CPU baseline; `value` is then the B = --cpu-rays figure")
    p.add_argument("--cpu-full", action="store_true", help="(default since round 6; kept for old command lines)")
    p.add_argument("--cpu-budget-s", type=float, default=330.0,
                   help="wall-clock cap of the whole CPU baseline: a shape that would exceed it stops early and reports timed_steps < requested")
    p.add_argument("--converge-steps", type=int, default=2000, help="training budget of the `converge` record")
    p.add_argument("--converge-views", type=int, default=2, help="held-out 800x800 views rendered for eval PSNR")
    # sizes of the other records; the defaults are the BASELINE configs, the dry run passes small ones
    p.add_argument("--strong-rays", type=int, default=512)
    p.add_argument("--octree-cams", type=int, default=4, help="views of the `octree` record (scripts/octree_bench.py uses 8)")
    p.add_argument("--grid-reso", type=int, default=512)
    p.add_argument("--eval-step", type=int, default=105,
                   help="render_fwd / grid512 are evaluated on the parameters after exactly this many train steps")
    p.add_argument("--image-factor", type=int, default=0, help="0 = 800x800 views (datasets.Synthetic)")
    p.add_argument("--sparsity-npoints", type=int, default=None, help="dry run only (preset value otherwise)")
    p.add_argument("--backend", choices=["nccl", "gloo"], default="nccl", help="gloo = CPU dry run (tests only)")
    p.add_argument("--share-gpu", action="store_true",
                   help="with --backend gloo: every rank drives cuda:0 with the REAL kernels (functional check of the N > 1 "
                        "flow on a 1-GPU box: collectives through gloo, timings meaningless)")
    return p.parse_args(argv)


def launch_command(n_gpus, argv, port):
    """The torch.distributed.run command bench.py re-executes itself through for --gpus N > 1."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def self_launch(a, argv):
    """`python bench.py --gpus N` without a launcher: one rank per GPU over RCCL."""
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < a.gpus and not a.share_gpu:
        raise SystemExit(f"bench.py --gpus {a.gpus}: only {have} ROCm device(s) visible")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC (RCCL between processes)
    return subprocess.call(launch_command(a.gpus, argv, port), env=env)


def flags_for(a, preset, batch, **over):
    from plenoctree_amd.nerf_sh.nerf import utils
    args = utils.define_flags().parse_args([])
    args.config = preset
    utils.update_flags(args)
    args.dataset = "synthetic"
    args.batch_size = batch
    args.factor = a.image_factor          # 0: 800 x 800
    args.train_dir = "/tmp/pxo_bench"
    args.skip_zero_rows = False           # every throughput record runs the DENSE reverse pass (the reference's value_and_grad)
    if a.sparsity_npoints is not None:
        args.sparsity_npoints = a.sparsity_npoints
    for k, v in over.items():
        setattr(args, k, v)
    return args


def cpu_baseline(a, args_ns, device, long_legs=True):
    """The oracle (CPU restatement of the reference graph, not JAX) timed on this host's cores by BASELINE.md section 3's protocol:
    3 warm-up + 10 timed train steps (forward + backward + Adam, float32), rays/s = B x steps/s (nerf_sh/train.py:224), at
    configs[0]'s shapes -- B = 1024 with 64 coarse samples only and with 64 + 128 samples (+ 10k sparsity points) -- and at
    configs[1]'s B = 4096 (the headline's shape, ~3 minutes of CPU; --no-cpu-full leaves it out).  `value` is the B = 4096,
    64 + 128 figure (the B = 1024 one without it).  Thread count: torch's intra-op pool oversubscribes badly on many-core
    hosts, so a 32-ray probe picks among a few candidates; BASELINE.md section 3 says `nproc`, so the 64 + 128 shape at B = 1024
    is ALSO timed with every hardware thread (ONE step, no warm-up: 85 s on a 256-thread host, where start-up effects are noise)
    and reported next to the probed-best figure.  The whole
    leg is capped at --cpu-budget-s: a shape that runs out of budget stops early and says so (timed_steps < requested)."""
    import platform
    from oracle import nerf_oracle as O
    from plenoctree_amd.nerf_sh.nerf import datasets
    gen = torch.Generator().manual_seed(0)

    def cfg_for(fine):
        return O.Cfg(sh_deg=args_ns.sh_deg, near=args_ns.near, far=args_ns.far, sparsity_npoints=args_ns.sparsity_npoints,
                     sparsity_length=args_ns.sparsity_length, sparsity_radius=args_ns.sparsity_radius,
                     num_fine_samples=128 if fine else 0)

    def host_batches(n_rays):
        # the same feeder as the GPU legs (batches are drawn on the device and copied to the host before the clock starts)
        ds_dev = datasets.Synthetic("train", args_ns, device, batch_size=n_rays)

        class _HostBatches:
            def __next__(self):
                b = next(ds_dev)
                return {"rays": type(b["rays"])(*[t.cpu() for t in b["rays"]]), "pixels": b["pixels"].cpu()}
        return _HostBatches()

    def step_once(cfg, flat, m, v, step, batch):
        rays = O.Rays(*batch["rays"])
        n = rays.origins.shape[0]
        t_rand = torch.rand(n, 64, generator=gen)
        u = torch.rand(n, 128, generator=gen) if cfg.num_fine_samples > 0 else None
        sp = (torch.rand(cfg.sparsity_npoints, 3, generator=gen) * 2 - 1) * cfg.sparsity_radius
        t0 = time.perf_counter()
        out = O.train_step(flat, m, v, step, rays, batch["pixels"], cfg, t_rand, u, sp, 5e-4)
        return out[:3], time.perf_counter() - t0

    if a.cpu_steps < 1 or a.cpu_warmup < 0:
        raise SystemExit("bench.py: --cpu-steps must be >= 1 and --cpu-warmup >= 0")
    t_leg = time.perf_counter()
    ncpu = os.cpu_count() or 1
    cfg = cfg_for(True)
    flat = O.flatten_params(O.init_params(cfg))
    m = torch.zeros_like(flat); v = torch.zeros_like(flat)
    small = {k: (type(val)(*[t[:32] for t in val]) if k == "rays" else val[:32]) for k, val in next(host_batches(32)).items()}
    best, cores = None, 1
    for c in sorted({min(ncpu, x) for x in (8, 16, 32, 64, 128)}):
        torch.set_num_threads(c)
        step_once(cfg, flat, m, v, 0, small)
        _, dt = step_once(cfg, flat, m, v, 0, small)
        if best is None or dt < best:
            best, cores = dt, c
    torch.set_num_threads(cores)

    def protocol(n_rays, fine, warm=a.cpu_warmup, timed=a.cpu_steps, threads=None):
        torch.set_num_threads(threads or cores)
        cfg = cfg_for(fine)
        flat = O.flatten_params(O.init_params(cfg))
        m = torch.zeros_like(flat); v = torch.zeros_like(flat)
        ds = host_batches(n_rays)
        times, last = [], 0.0
        for step in range(warm + timed):
            # the budget: never start a step that (at the last step's duration) would end past the cap, once one step is timed
            if times and time.perf_counter() - t_leg + last > a.cpu_budget_s:
                break
            (flat, m, v), last = step_once(cfg, flat, m, v, step, next(ds))
            if step >= warm:
                times.append(last)
        torch.set_num_threads(cores)
        if not times:        # out of budget during the warm-up: the last warm-up step is the only measurement there is
            times = [last]
        return {"rays_per_step": n_rays, "samples": "64+128" if fine else "64", "threads": threads or cores, "warmup_steps": warm,
                "timed_steps": len(times), "timed_steps_requested": timed,
                "rays_per_s": n_rays * len(times) / sum(times), "s_per_step": sum(times) / len(times), "cpu_s": sum(times)}

    shapes = [protocol(a.cpu_rays, False), protocol(a.cpu_rays, True)]
    dropped = []
    full = long_legs and not a.no_cpu_full
    if full:
        shapes.append(protocol(4096, True))
    else:
        dropped.append("B=4096, 64+128 (configs[1]; ~3 min of CPU: left out " + ("by --no-cpu-full)" if long_legs else "at N > 1, see the N = 1 line)"))
    if long_legs and ncpu != cores:
        all_threads = protocol(a.cpu_rays, True, warm=0, timed=1, threads=ncpu)
    else:
        all_threads = dict(shapes[1])
        if ncpu != cores:
            all_threads["rays_per_s"] = None
            dropped.append("the nproc-thread step (left out at N > 1, see the N = 1 line)")
    cpu_model = ""
    try:
        with open("/proc/cpuinfo") as f:
            cpu_model = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "")
    except OSError:
        pass
    head = shapes[-1] if full else shapes[1]
    return {"value": head["rays_per_s"], "unit": "rays/s", "cores": torch.get_num_threads(), "kind": "port",
            "nproc": ncpu, "nproc_threads": {"rays_per_s": all_threads["rays_per_s"], "threads": ncpu if long_legs else all_threads["threads"], "rays_per_step": all_threads["rays_per_step"],
                                             "timed_steps": all_threads["timed_steps"],
                                             "note": "BASELINE.md section 3 says nproc threads: the same 64+128 shape with every hardware thread"}, "cpu_model": cpu_model or platform.processor(), "torch_version": torch.__version__,
            "threads_probed": cores, "protocol": f"BASELINE.md section 3: {a.cpu_warmup} warm-up + {a.cpu_steps} timed train steps, rays/s = B x steps/s",
            "shapes": shapes, "shapes_dropped": dropped,
            "sample": f"{head['timed_steps']} timed train steps (after {head['warmup_steps']} warm-up) of {head['rays_per_step']} rays x "
                      f"(64+128) samples + {args_ns.sparsity_npoints} sparsity points, oracle/nerf_oracle.py (torch-CPU f32 "
                      f"restatement, not JAX), {head['cpu_s']:.1f} s on rank 0's host cores"}


def hbm_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed PMC passes of this same command
    (profiles/hbm_traffic.json; rocprofv3 cannot run inside the timed process).  (None, None) if absent."""
    try:
        with open(os.path.join(ROOT, "profiles", "hbm_traffic.json")) as f:
            doc = json.load(f)
        k = doc["kernels"][kernel]
        return (k["hbm_read_bytes_per_launch"] + k["hbm_write_bytes_per_launch"],
                "profiles/hbm_traffic.json (" + doc.get("source", "rocprofv3 --pmc passes of this command, committed") +
                (", code at commit " + doc["commit"] if doc.get("commit") else "") + "); not measured inside this run")
    except Exception:
        return None, None


class Job:
    """Rank context shared by the legs of the benchmark."""

    def __init__(self, a):
        from plenoctree_amd import dist as pdist
        self.a = a
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.cuda = a.backend == "nccl" or a.share_gpu
        if self.cuda:
            dev_index = 0 if a.share_gpu else self.local_rank
            torch.cuda.set_device(dev_index)
            self.device = torch.device("cuda", dev_index)
        else:
            self.device = torch.device("cpu")
        self.dist = None
        self.ranks_seen = 1
        self.pdist = pdist
        self.exchange = self.world > 1 or a.force_dist       # do the legs other than strong512 exchange gradients?
        want_strong = self.cuda and not a.no_extras and "strong512" in a.extras.split(",")
        if self.exchange or want_strong:
            # the group (and with it RCCL's high-priority stream) is created BEFORE any kernel runs: created after the
            # headline leg, the same stream made the 512-ray step 26 % slower (4.55 vs 3.60 ms, profiles/r04h_late_group.txt)
            self.init_group()
            if self.cuda:
                self.pdist.exchange_stream(self.device)
            one = torch.ones(1, device=self.device)
            self.dist.all_reduce(one)
            self.ranks_seen = int(one.item())

    def init_group(self):
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(29533 + os.getpid() % 2000))
        kw = {"device_id": self.device, "pg_options": self.pdist.nccl_options()} if self.a.backend == "nccl" else {}
        dist_mod.init_process_group(self.a.backend, rank=self.rank, world_size=self.world, **kw)   # "nccl" = RCCL over xGMI
        self.dist = dist_mod

    def reducer(self, force=None):
        """The per-step gradient exchange (two buckets, dist.GradReducer): on with more than one rank or --force-dist, and in
        the strong512 leg (`force=True`: through RCCL even with one rank)."""
        force = self.exchange if force is None else force
        return self.pdist.GradReducer(self.comm(), self.device, force=bool(force) and self.dist is not None)

    def sync(self):
        if self.dist:
            self.dist.barrier()
        if self.cuda:
            torch.cuda.synchronize()

    def max_over_ranks(self, seconds):
        if not self.dist:
            return seconds
        t = torch.tensor([seconds], dtype=torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def comm(self):
        return self.pdist.Comm(self.world, self.rank, self.local_rank, self.a.backend if self.dist else None)


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


def run_train(job, preset, steps, warmup, per_gpu=None, snapshot_step=None, events=True, exchange=None, **flag_over):
    """W untimed + K timed train steps of `preset`; returns a dict with elapsed (max over ranks), kernels, stats.
    snapshot_step: a copy of the parameters after exactly that many steps from the fixed-seed initialisation is kept
    (taken inside the run if it gets that far, by untimed extra steps otherwise), so that the records evaluated on
    "a trained-ish network" do not depend on --steps / --warmup."""
    from plenoctree_amd import ops
    from plenoctree_amd.nerf_sh.nerf import datasets, models, utils
    a = job.a
    if per_gpu is None:
        per_gpu = a.batch if a.scaling == "weak" else a.batch // job.world
    args = flags_for(a, preset, per_gpu, **flag_over)
    model, params = models.construct_nerf(args, job.device)
    state = models.TrainState(model.cfg, params)
    if job.pdist.per_host_image(a.per_host_image, job.world):
        dataset = datasets.Synthetic("train", args, job.device, batch_size=per_gpu, seed=20201473, shard=(job.rank, job.world))
    else:
        dataset = datasets.Synthetic("train", args, job.device, batch_size=per_gpu, seed=20201473 + job.rank)
    reducer = job.reducer(exchange)
    snap = {}

    def one_step(step):
        batch = next(dataset)
        lr = utils.learning_rate_decay(step, args.lr_init, args.lr_final, args.max_steps)
        models.train_step(model, state, batch, lr, randomized=True, seed=(step << 8) | job.rank,
                          world_size=job.world, reducer=reducer)
        if state.step == snapshot_step:
            snap["params"] = state.params.clone()          # 4 MB device copy

    for s in range(warmup):
        one_step(s)
    job.sync()
    # Inside the timed region only the dominant kernel (mlp_fwd, the `roofline` leg) is bracketed by HIP events: an event
    # record is a barrier packet between two kernels that would otherwise dispatch back to back (~5 us each; with all
    # four tags on, 16 brackets = 0.13 ms of a 3.7 ms step at 512 rays -- profiles/r04a_events_ab.txt).
    ops.profile_enable(tags=[] if (a.no_kernel_events or not events) else [ops.PROF_MLP_FWD])
    t0 = time.perf_counter()
    for s in range(warmup, warmup + steps):
        one_step(s)
    job.sync()
    elapsed = time.perf_counter() - t0
    ops.profile_enable(False)
    elapsed = job.max_over_ranks(elapsed)
    deg = model.cfg.sh_deg
    stats = dict(zip(utils.Stats._fields, state.stats.cpu().tolist()))
    dom = read_kernels(ops, deg)
    kernels = dom
    if events and not a.no_kernel_events and job.cuda:
        # the table of the other kernels: a few more steps, untimed, every tag bracketed (mlp_fwd's row stays the timed one)
        extra = min(10, max(steps, 1))
        ops.profile_enable(True)
        for s in range(warmup + steps, warmup + steps + extra):
            one_step(s)
        job.sync()
        ops.profile_enable(False)
        kernels = dom + [k for k in read_kernels(ops, deg) if k["kernel"] != "mlp_fwd_kernel"]
        for k in kernels[len(dom):]:
            k["from"] = f"{extra} untimed steps after the timed region"
    out = {"elapsed": elapsed, "per_gpu": per_gpu, "deg": deg, "kernels": kernels,
           "stats": stats, "args": args,
           "model": model, "state": state, "dataset": dataset, "one_step": one_step,
           "collectives_per_step": 2 if reducer.active else 0}
    if snapshot_step is not None:
        for s in range(state.step, snapshot_step):          # untimed: a short run did not get there
            one_step(s)
        out["eval_state"] = models.TrainState(model.cfg, snap["params"])
        out["eval_step"] = snapshot_step
    return out


def x6_state_twin(tr):
    """(model, state) at the SAME parameters / Adam moments / step with PxoCfg.mlp_precision = bf16x6 (its own weight images)."""
    from plenoctree_amd.nerf_sh.nerf import models
    cfg = type(tr["model"].cfg).from_buffer_copy(tr["model"].cfg)
    cfg.mlp_precision = 2
    st = models.TrainState(cfg, tr["state"].params.clone(), step=tr["state"].step)
    st.m.copy_(tr["state"].m); st.v.copy_(tr["state"].v)
    return models.NerfModel(cfg), st


def eval_views(job, a, tr):
    """render_image of `--converge-views` held-out views of the TEST split at tr's current parameters, deterministic sampling;
    -> ([psnr per view], seconds (max over ranks), the test set)."""
    from plenoctree_amd.nerf_sh.nerf import datasets, utils
    test = datasets.Synthetic("test", tr["args"], job.device)
    model, state = tr["model"], tr["state"]
    comm = job.comm()
    n_views = min(a.converge_views, test.size)
    psnrs = []
    job.sync()
    t1 = time.perf_counter()
    for i in range(n_views):
        ex = test.get_image(i * (test.size // n_views))
        rgb, _, _ = utils.render_image(lambda r: model.apply(state, r, False), ex["rays"], chunk=a.batch * job.world,
                                       world_size=job.world, rank=job.rank, gather=comm.all_gather_cat)
        psnrs.append(utils.compute_psnr(((rgb - ex["pixels"]) ** 2).mean().item()))
    job.sync()
    return psnrs, job.max_over_ranks(time.perf_counter() - t1), test


def run_converge(job, a):
    """The second half of BASELINE.json's metric: eval PSNR.  A fixed training budget from the fixed-seed initialisation
    (`--converge-steps` steps of `--batch` rays per GPU, the headline's step), then render_image of held-out views of the
    TEST split with deterministic sampling (nerf_sh/eval.py:57, nerf_sh/train.py:245-268, utils.py:331-381) and
    compute_psnr (utils.py:384-393) against their ground truth."""
    from plenoctree_amd.nerf_sh.nerf import datasets, utils
    job.sync()
    t0 = time.perf_counter()
    tr = run_train(job, a.preset, a.converge_steps, 0, events=False)
    t_train = tr["elapsed"]
    model, state = tr["model"], tr["state"]
    psnrs, t_render, test = eval_views(job, a, tr)
    # At this trained state: the same steps with the dense reverse pass and with PxoCfg.skip_zero_rows (sample rows whose
    # upstream gradient is exactly zero -- empty space, occluded samples, background rays -- left out of backward(data) and
    # the weight-gradient GEMMs in 16-row chunks; bit-identical gradients, tests/test_gpu_parity.py).  Not the headline:
    # the saving is a property of the scene, here 78 % background pixels.
    from plenoctree_amd import ops

    def timed_steps(first, n):
        job.sync()
        t = time.perf_counter()
        for s_ in range(first, first + n):
            tr["one_step"](s_)
        job.sync()
        return tr["per_gpu"] * job.world * n / job.max_over_ranks(time.perf_counter() - t)

    k = 50 if job.cuda else 1
    s0 = a.converge_steps
    dense_rps = timed_steps(s0, k)
    sparse = {"dense_rays_per_s": dense_rps}
    if job.cuda:
        model.cfg.skip_zero_rows = 1
        timed_steps(s0 + k, 3)
        sparse["skip_zero_rows_rays_per_s"] = timed_steps(s0 + k + 3, k)
        live, total = ops.train_backward_work(model.cfg, tr["per_gpu"], state._ws)
        model.cfg.skip_zero_rows = 0
        sparse.update(live_chunk_fraction=live / max(total, 1), steps_each=k, after_steps=s0,
                      note="opt-in PxoCfg.skip_zero_rows: rows with an exactly zero upstream gradient skipped in 16-row chunks; "
                           "gradients bit-identical to the dense pass; scene-dependent, not part of any other record")
        if "bf16x6" in a.extras.split(",") and not a.no_extras:
            # the same trained state through the opt-in bf16x6 kernels (its own weight images), dense and skipping
            from plenoctree_amd.nerf_sh.nerf import models, utils
            m6, s6 = x6_state_twin(tr)
            reducer = job.reducer()

            def steps6(first, n):
                job.sync()
                t = time.perf_counter()
                for s_ in range(first, first + n):
                    lr = utils.learning_rate_decay(s_, tr["args"].lr_init, tr["args"].lr_final, tr["args"].max_steps)
                    models.train_step(m6, s6, next(tr["dataset"]), lr, randomized=True, seed=(s_ << 8) | job.rank,
                                      world_size=job.world, reducer=reducer)
                job.sync()
                return tr["per_gpu"] * job.world * n / job.max_over_ranks(time.perf_counter() - t)
            steps6(s0, 3)
            sparse["bf16x6_dense_rays_per_s"] = steps6(s0 + 3, k)
            m6.cfg.skip_zero_rows = 1
            steps6(s0 + 3 + k, 3)
            sparse["bf16x6_skip_zero_rows_rays_per_s"] = steps6(s0 + 6 + k, k)
            m6, s6 = None, None
    return {"eval_psnr": sum(psnrs) / len(psnrs), "sparse_backward": sparse, "eval_psnr_per_view": psnrs, "views": len(psnrs),
            "view_size": [test.h, test.w], "train_steps": a.converge_steps, "rays_per_step": tr["per_gpu"] * job.world,
            "train_s": t_train, "train_rays_per_s": tr["per_gpu"] * job.world * a.converge_steps / t_train,
            "train_psnr_last_batch": tr["stats"]["psnr"], "render_s": t_render,
            "render_rays_per_s": len(psnrs) * test.h * test.w / t_render, "wall_s": time.perf_counter() - t0,
            "data": "synthetic analytic scene (three shaded spheres, white background): 100 train / 200 test poses, "
                    "datasets.Synthetic; seed-fixed initialisation and batches",
            "sampling": "train randomized (Philox), eval deterministic"}


def run_strong(job, a):
    """The strong-scaling shape of BASELINE configs[2]: the reference's global batch of 4096 rays over 8 GPUs = 512
    rays per GPU per step (train.py:117-118: pmean per step).  Measured on this job's GPUs with 512 rays each and
    the per-step collectives issued through RCCL even with one rank (two buckets, dist.GradReducer), so the number is
    what one GPU of an 8-GPU strong-scaling run does before the wire time of the exchange."""
    err = None
    if job.dist is None:                              # (the group is created in Job.__init__ whenever this leg is wanted)
        err = "no process group"
    rays = a.strong_rays
    k = max(40, a.steps) if job.cuda else a.steps
    t = run_train(job, a.preset, k, 5 if job.cuda else 1, per_gpu=rays, exchange=True)
    rec = {"value": rays * job.world * k / t["elapsed"], "unit": "rays/s", "rays_per_gpu": rays, "steps": k,
           "ms_per_step": 1e3 * t["elapsed"] / k, "collectives_per_step": t["collectives_per_step"],
           "frac": rays * k / t["elapsed"] * FLOP_TRAIN_PER_RAY[t["deg"]] / (PEAK_F32_MFMA_TFLOPS * 1e12),
           "kernels": [{"kernel": e["kernel"], "avg_ms": e["avg_ms"], "tflops": e.get("tflops")} for e in t["kernels"]],
           "note": "per-GPU work of the 8-GPU strong-scaling run of the reference's 4096-ray batch; 10k sparsity points "
                   "per GPU per step (train.py:78-80) are not counted as rays"}
    if err:
        rec["rccl_init_error"] = err
    return rec


def run_coarse64(job, a):
    """BASELINE configs[0] shape on the GPU ("1k rays x 64 samples", north_star's "800x800x64-sample batches"): the
    coarse level only (num_fine_samples = 0; sparsity points ride with the coarse pass), 1024 and `--batch` rays."""
    out = {}
    for rays in sorted({min(1024, a.batch), a.batch}):
        k = max(20, a.steps // 2) if job.cuda else a.steps
        t = run_train(job, a.preset, k, 3 if job.cuda else 1, per_gpu=rays, num_fine_samples=0)
        v = rays * job.world * k / t["elapsed"]
        out[f"rays{rays}"] = {"value": v, "unit": "rays/s", "steps": k, "ms_per_step": 1e3 * t["elapsed"] / k,
                              "frac": v / job.world * FLOP_TRAIN_PER_RAY_COARSE_ONLY[t["deg"]] / (PEAK_F32_MFMA_TFLOPS * 1e12)}
    out["workload"] = "64 coarse samples per ray, no fine level (SURVEY 8d: 189.2 MFLOP per ray), 10k sparsity points"
    return out


def run_x6(job, a, f32_head):
    """Opt-in PxoCfg.mlp_precision = bf16x6 (csrc/mlp_x6_kernels.hip): the headline's step with the fused MLP forward (saved
    tensors) and backward(data) evaluated as six bf16 partial products per float32 product, float32 accumulation, everything
    that leaves the kernels float32; the 256x256 weight-gradient products take their float32 operands through the same split
    (csrc/wgrad_x6_kernels.hip), the skinny ones stay on the float32 pipe.  Same batches, same seeds, dense reverse pass."""
    k = max(20, a.steps // 2) if job.cuda else a.steps
    t = run_train(job, a.preset, k, 3 if job.cuda else 1, mlp_precision="bf16x6")
    v = t["per_gpu"] * job.world * k / t["elapsed"]
    rec = {"value": v, "unit": "rays/s", "dtype": "f32 (bf16x6 emulated)", "steps": k, "ms_per_step": 1e3 * t["elapsed"] / k,
           "vs_f32_headline": v / f32_head if f32_head else None,
           "kernels": [{"kernel": e["kernel"] + (" (bf16x6)" if e["kernel"].startswith("mlp_") else ""), "avg_ms": e["avg_ms"],
                        "equivalent_f32_tflops": e.get("tflops")} for e in t["kernels"]],
           "final_stats": t["stats"],
           "note": "opt-in, float32-accurate: x = x1 + x2 + x3 exactly (bf16 each), the six products of order <= 2^-16 on "
                   "v_mfma_f32_32x32x16_bf16, leading product and corrections in separate float32 accumulators; per GEMM at least "
                   "as close to float64 as the float32-MFMA kernels (the 256x256 weight-gradient products included) and held to every bound of the float32 path "
                   "(tests/test_gpu_x6.py, test_gpu_fullsize.py, test_gpu_trained_state.py, test_gpu_reference_fixtures.py); "
                   "`equivalent_f32_tflops` = the float32 path's algorithmic FLOP over this kernel's time (NOT a bf16 rate)"}
    if "bf16x6_converge" in a.extras.split(","):
        # opt-in (not in the default line: +45 s): the `converge` record's run -- the same budget, initialisation, batches and seeds --
        # trained END TO END in bf16x6, and its eval PSNR on the same held-out views
        job.sync()
        t0 = time.perf_counter()
        tr = run_train(job, a.preset, a.converge_steps, 0, events=False, mlp_precision="bf16x6")
        psnrs, t_render, _ = eval_views(job, a, tr)
        rec["converge"] = {"eval_psnr": sum(psnrs) / len(psnrs), "eval_psnr_per_view": psnrs, "train_steps": a.converge_steps,
                           "rays_per_step": tr["per_gpu"] * job.world, "train_s": tr["elapsed"],
                           "train_rays_per_s": tr["per_gpu"] * job.world * a.converge_steps / tr["elapsed"],
                           "train_psnr_last_batch": tr["stats"]["psnr"], "render_s": t_render, "wall_s": time.perf_counter() - t0}
    return rec


def split_precision_twin(tr, precision=1):
    """Model/state pair with the same parameters and PxoCfg.mlp_precision = bf16x3 (1, opt-in inference path) or bf16x6 (2)."""
    from plenoctree_amd.nerf_sh.nerf import models
    cfg = type(tr["model"].cfg).from_buffer_copy(tr["model"].cfg)
    cfg.mlp_precision = precision
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
    if not job.cuda:
        iters = 2
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


def run_grid(job, tr, stages_after_grid=True):
    """BASELINE configs[4]: step 1 of octree.extraction at init_grid_depth 8 (octree/extraction.py:288-352):
    sigma of MLP_1 on the 512^3 grid (x-slabs sharded over the ranks + all-gather), the weight mask over the 100
    training views (cameras sharded + max-all-reduce) and the tree build."""
    from plenoctree_amd import octree_ops as oops
    from plenoctree_amd.octree import extraction
    from plenoctree_amd.octree.svox import N3Tree
    model, state, dataset = tr["model"], tr["eval_state"], tr["dataset"]
    comm = job.comm()
    reso, center, radius = job.a.grid_reso, [0.0, 0.0, 0.0], [1.5, 1.5, 1.5]
    if model.cfg.mlp_precision == 0:
        state.repack(need_bwd=False)
    extraction.grid_sigma(model, state, min(64, reso), center, radius, comm)       # warm-up (small grid)
    job.sync()
    t0 = time.perf_counter()
    sig = extraction.grid_sigma(model, state, reso, center, radius, comm)
    job.sync()
    t_grid = job.max_over_ranks(time.perf_counter() - t0)
    if stages_after_grid is False:
        del sig
        tflops = reso ** 3 * FLOP_SIGMA_PER_POINT / t_grid / 1e12
        return {"points": reso ** 3, "grid_ms": 1e3 * t_grid, "equivalent_f32_tflops": tflops}
    tree = N3Tree(N=2, data_dim=1 + 3 * (tr["deg"] + 1) ** 2, init_refine=0, depth_limit=int(math.log2(reso)) - 1,
                  radius=radius, center=center, data_format=f"SH{(tr['deg'] + 1) ** 2}", map_location=job.device)
    t0 = time.perf_counter()
    weights = extraction.calculate_grid_weights(dataset, sig, reso, tree.invradius, tree.offset, 1e-4, comm)
    job.sync()
    t_weight = job.max_over_ranks(time.perf_counter() - t0)
    # an untrained network has no surfaces: threshold sigma at its 97th percentile, i.e. a mask of ~4 M voxels as a
    # trained scene leaves (the build time depends on how many voxels are set, not on which)
    thr = float(torch.quantile(sig[::4099 if reso >= 64 else 1].float(), 0.97))
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


def run_octree(job, a):
    """SURVEY 8(f) rows 2-3 in the driver-run line: the PlenOctree-side kernels at the reference's sizes (512^3 grid, 800 x 800
    views, SH16) on the analytic scene of scripts/octree_bench.py -- weight mask per camera (octree/extraction.py:181-214),
    tree build, step-2 sampling (:369), VolumeRenderer.render_persp exact and early-stop per view (octree/optimization.py:195),
    its backward per view (:216) and one SGD step over the tree -- each with its counted algorithmic bytes, the fraction of
    6.3 TB/s (and of the 8 TB/s specification) they are moved at, and the HBM bytes of the committed PMC passes.  Every rank
    measures its own GPU (nothing here is sharded); rank 0's numbers are reported."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import octree_bench
    job.sync()
    t0 = time.perf_counter()
    m = octree_bench.measure(octree_bench.defaults(cams=a.octree_cams, reps=1))
    job.sync()
    keep = ("basis_dim", "reso", "image", "cams", "step_size", "grid_weight_render_ms_per_cam", "grid_weight_roofline", "mask_voxels",
            "tree_build_ms", "n_internal", "sample_cells_ms", "sample_points", "render_exact_ms_per_image", "render_exact_roofline",
            "render_fast_ms_per_image", "render_fast_roofline", "render_bwd_ms_per_image", "render_bwd_roofline",
            "render_bwd_reusing_fwd_ms_per_image", "render_bwd_reusing_fwd_roofline", "sgd_ms", "sgd_roofline", "tree_data_MB")
    rec = {k: m[k] for k in keep if k in m}
    for k, v in rec.items():                      # the per-image work counts stay in profiles/*_octree_bench.json
        if isinstance(v, dict) and "per_image" in v:
            v.pop("per_image")
    rec["wall_s"] = time.perf_counter() - t0
    rec["scene"] = "scripts/octree_bench.py: three fuzzy spheres on the 512^3 grid, random SH16 leaves; same command as profiles/*_octree_bench.json"
    return rec


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    a = parse(argv)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(a, argv))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus != world:
        raise SystemExit(f"bench.py --gpus {a.gpus} but WORLD_SIZE={world}")
    if (a.backend == "nccl" or a.share_gpu) and not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU; the HIP path has no CPU fallback")
    if a.share_gpu and a.backend != "gloo":
        raise SystemExit("bench.py --share-gpu needs --backend gloo (RCCL refuses two ranks on one device)")
    job = Job(a)
    rank = job.rank

    if job.cuda:
        from plenoctree_amd import build
        if rank == 0:
            build.build(verbose=False)
        if job.dist:
            job.dist.barrier()

    tuning = {}
    if a.tune:
        from plenoctree_amd import ops
        knobs = {"tile_sched": ops.TUNE_TILE_SCHED, "wgrad_ranges": ops.TUNE_WGRAD_RANGES,
                 "wgrad_skinny_ranges": ops.TUNE_WGRAD_SKINNY_RANGES, "coarse_stream": ops.TUNE_COARSE_REVERSE_STREAM}
        for kv in a.tune.split(","):
            k, _, v = kv.partition("=")
            if k not in knobs:
                raise SystemExit(f"bench.py --tune: unknown knob {k!r} (have {sorted(knobs)})")
            if job.cuda:
                ops.set_tuning(knobs[k], int(v))
            tuning[k] = int(v)

    want = [] if a.no_extras else [e for e in a.extras.split(",") if e]
    unknown = [e for e in want if e not in ALL_EXTRAS + OPT_IN_EXTRAS]
    if unknown:
        raise SystemExit(f"bench.py --extras: unknown record(s) {unknown}")
    need_snapshot = any(e in want for e in ("render_fwd", "grid512", "bf16x3", "bf16x6"))
    tr = run_train(job, a.preset, a.steps, a.warmup, snapshot_step=a.eval_step if need_snapshot else None)
    extras = {}
    if "strong512" in want:
        extras["strong512"] = run_strong(job, a)
    if "render_fwd" in want:
        extras["render_fwd"] = run_render(job, tr)
    if "grid512" in want:
        extras["grid512"] = run_grid(job, tr)
    if "octree" in want and job.cuda:
        extras["octree"] = run_octree(job, a)
    if "bf16x6" in want and job.cuda:
        extras["opt_in_bf16x6_training"] = run_x6(job, a, tr["per_gpu"] * world * a.steps / tr["elapsed"])
        if job.cuda:
            # the forward-only entry points in the same precision: eval rendering and the 512^3 sigma grid
            twin = split_precision_twin(tr, 2)
            r6, g6 = run_render(job, twin), run_grid(job, twin, stages_after_grid=False)
            extras["opt_in_bf16x6_training"]["inference"] = {
                "render_fwd_rays_per_s": r6["value"], "render_fwd_ms_per_call": r6["ms_per_call"],
                "grid512_ms": g6["grid_ms"], "grid512_equivalent_f32_tflops": g6["equivalent_f32_tflops"]}
            twin = None
    if "bf16x3" in want:
        # opt-in inference precision (NOT the headline, NOT used in training): products as 3 bf16 MFMAs, f32 accumulate
        twin = split_precision_twin(tr)
        r3, g3 = run_render(job, twin), run_grid(job, twin, stages_after_grid=False)
        extras["opt_in_bf16x3_inference"] = {
            "note": "PxoCfg.mlp_precision = bf16x3: forward-only, |dPSNR| vs the f64 oracle <= 1e-4 dB (tests/test_gpu_x3.py); "
                    "training and the headline stay float32",
            "render_fwd_rays_per_s": r3["value"], "render_fwd_ms_per_call": r3["ms_per_call"],
            "grid512_ms": g3["grid_ms"], "grid512_equivalent_f32_tflops": g3["equivalent_f32_tflops"]}
        twin = None
    head = {k: tr[k] for k in ("per_gpu", "deg", "kernels", "elapsed", "stats", "args", "collectives_per_step")}
    tr = None                                  # release the headline workspace (19 GB) before the other presets
    if job.cuda:
        torch.cuda.empty_cache()
    if "coarse64" in want:
        extras["coarse64"] = run_coarse64(job, a)
    if "tt_sh25" in want:
        other = "tt" if a.preset == "blender" else "blender"
        k2 = max(10, a.steps // 4) if job.cuda else a.steps
        t2 = run_train(job, other, k2, 3 if job.cuda else 1)
        v2 = t2["per_gpu"] * world * k2 / t2["elapsed"]
        extras["tt_sh25" if other == "tt" else "blender_sh16"] = {
            "value": v2, "unit": "rays/s", "steps": k2, "ms_per_step": 1e3 * t2["elapsed"] / k2,
            "frac": v2 / world * FLOP_TRAIN_PER_RAY[t2["deg"]] / (PEAK_F32_MFMA_TFLOPS * 1e12),
            "mlp_fwd_tflops": t2["kernels"][0]["tflops"] if t2["kernels"] else None,
            "workload": "nerf_sh/config/tt.yaml: SH25, near 0, far 4, sparsity_length 0.2, sparsity_radius 5"
                        if other == "tt" else "nerf_sh/config/blender.yaml"}
        t2 = None
        if job.cuda:
            torch.cuda.empty_cache()
    if "converge" in want:
        extras["converge"] = run_converge(job, a)

    # the CPU baseline is timed on rank 0's host cores.  Its long legs (B = 4096, every hardware thread: ~4 of its 5 minutes
    # since round 6) run at N = 1 only: at N > 1 the other ranks sit in the closing barrier's collective meanwhile, next to its
    # watchdog, so they get the two B = 1024 shapes (~1 minute, as in round 5) and the line says which were left out
    cpu = None
    if rank == 0 and not a.no_cpu_baseline:
        cpu = cpu_baseline(a, head["args"], job.device, long_legs=world == 1)
    if rank == 0:
        per_gpu, deg, kernels = head["per_gpu"], head["deg"], head["kernels"]
        elapsed = head["elapsed"]
        value = per_gpu * world * a.steps / elapsed
        dom = kernels[0] if kernels else None            # mlp_fwd_kernel: largest single launch of the step
        roofline = None
        if dom:
            traffic, source = hbm_traffic("mlp_fwd_kernel")
            roofline = {"bound": "mfma", "kernel": dom["kernel"], "achieved": dom["tflops"],
                        "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": dom["tflops"] / PEAK_F32_MFMA_TFLOPS,
                        "avg_launch_ms": dom["avg_ms"], "launches": dom["launches"],
                        "flop_per_launch": dom["rows_per_launch"] * FLOP_FWD_PER_ROW[deg],
                        "traffic": traffic, "traffic_source": source}
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
            "batch_sampler": ("one image per step, its pixels sharded over the ranks (the reference on one host: datasets.py:159-166 + "
                              "utils.shard)" if job.pdist.per_host_image(a.per_host_image, world) else
                              "every rank its own image per step (the reference's multi-host sampler, train.py:128)") if world > 1 else
                             "one image per step (datasets.py:159-166)",
            "nccl_ranks_seen": job.ranks_seen if (job.exchange or world > 1) else 1, "collectives_per_step": head["collectives_per_step"],
            "step_mfma_frac": value / world * FLOP_TRAIN_PER_RAY[deg] / (PEAK_F32_MFMA_TFLOPS * 1e12),
            "final_stats": head["stats"],
            "roofline": roofline, "kernels": kernels,
        }
        if tuning:
            out["tuning"] = tuning
        if "converge" in extras:
            out["eval_psnr"] = extras["converge"]["eval_psnr"]      # the metric's second half, next to `value`
        out.update(extras)
        if cpu is not None:
            out["cpu_baseline"] = cpu
        if not job.cuda:
            out["dry_run"] = "gloo ranks on CPU with the test harness's stand-ins: control flow only, numbers meaningless"
        elif a.share_gpu:
            out["dry_run"] = (f"{world} gloo ranks sharing ONE GPU with the real kernels: functional check of the N > 1 flow, "
                              "timings meaningless")
        print(json.dumps(out), flush=True)
        # RCCL keeps its version banner in the C stdio buffer until the process exits: whatever libraries still flush to
        # fd 1 after this point goes to stderr, so that the JSON line above stays the only line on stdout
        sys.stdout.flush()
        if job.cuda:
            os.dup2(2, 1)
    if job.dist:
        job.dist.barrier()
        job.dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
