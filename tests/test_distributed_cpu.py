"""N > 1 host logic on CPU: two gloo ranks run the product's `models.train_step`,
`utils.render_image` and `octree.extraction.grid_sigma` with the HIP entry points monkeypatched by
oracle-backed stand-ins (test infrastructure only -- the product path has no CPU fallback).
Checks pmean semantics (nerf_sh/train.py:117-118): 2 x B/2 rays == 1 x B rays."""
import os
import socket
import sys
import tempfile

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import nerf_oracle as O  # noqa: E402

B_GLOBAL, N_SP, STEPS = 8, 32, 2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _oracle_cfg():
    return O.Cfg(sparsity_npoints=N_SP)


def _problem():
    cfg = _oracle_cfg()
    gen = torch.Generator().manual_seed(99)
    flat = O.flatten_params(O.init_params(cfg, seed=5))
    flat = flat + 0.02 * torch.randn(flat.shape, generator=gen)
    n = flat.numel() // 2
    b8 = sum(fi * fo + fo for fi, fo in O.layer_shapes(cfg)[:8]) + 256
    flat[b8] += 1.0; flat[n + b8] += 1.0
    cam = torch.randn(B_GLOBAL, 3, generator=gen); cam = 4 * cam / cam.norm(dim=-1, keepdim=True)
    d = -cam / 4 + 0.05 * torch.randn(B_GLOBAL, 3, generator=gen)
    rays = O.Rays(cam, d, d / d.norm(dim=-1, keepdim=True))
    px = torch.rand(B_GLOBAL, 3, generator=gen)
    rnd = [(torch.rand(B_GLOBAL, 64, generator=gen), torch.rand(B_GLOBAL, 128, generator=gen),
            (torch.rand(N_SP, 3, generator=gen) * 2 - 1) * 1.5) for _ in range(STEPS)]
    return cfg, flat, rays, px, rnd


def _patch_ops(cfg):
    """Oracle-backed stand-ins for the HIP entry points used by models.train_step."""
    from plenoctree_amd import ops

    def pack_weights(pcfg, mlp_params, f=None, b=None, need_bwd=True):
        return mlp_params, mlp_params

    def train_fwd_bwd(pcfg, params, packed, o, d, v, px, grads, stats, ws, randomized=True, t_rand=None, u=None,
                      sp_points=None, seed=0, grads0_ready=None):
        assert grads0_ready is None          # no device events on the CPU: the reducer then makes its two calls in order
        _, st, g = O.loss_and_grad(params, O.Rays(o, d, v), px, cfg, t_rand, u, sp_points)
        grads.copy_(g)
        stats.copy_(torch.stack([st[k] for k in ("loss", "psnr", "loss_c", "loss_sp", "psnr_c", "weight_l2")]))

    def adam_step(params, m, v, grads, lr, step, grad_scale=1.0):
        p, m2, v2 = O.adam_update(params, m, v, grads * grad_scale, lr, step)
        params.copy_(p); m.copy_(m2); v.copy_(v2)

    def adam_pack_step(pcfg, params, m, v, grads, lr, step, packed, grad_scale=1.0):
        adam_step(params, m, v, grads, lr, step, grad_scale)     # the stand-in images alias the parameters

    ops.pack_weights = pack_weights
    ops.train_fwd_bwd = train_fwd_bwd
    ops.adam_step = adam_step
    ops.adam_pack_step = adam_pack_step
    ops.train_workspace_bytes = lambda pcfg, B: 16
    ops.render_workspace_bytes = lambda pcfg, B: 16

    def grid_sigma(pcfg, packed, reso, x0, x1, offset, scale, out=None):
        ix = torch.arange(x0, x1, dtype=torch.float32)[:, None, None]
        iy = torch.arange(reso, dtype=torch.float32)[None, :, None]
        iz = torch.arange(reso, dtype=torch.float32)[None, None, :]
        val = (ix * 10000 + iy * 100 + iz).reshape(-1)
        out.copy_(val)
        return out

    ops.grid_sigma = grid_sigma


def _worker(rank, world, port, outdir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from plenoctree_amd import dist, ops
    from plenoctree_amd.nerf_sh.nerf import models, utils
    from plenoctree_amd.octree import extraction
    comm = dist.init_from_env(backend="gloo")
    assert comm.world == world and comm.rank == rank
    cfg, flat, rays, px, rnd = _problem()
    _patch_ops(cfg)
    pcfg = ops.make_cfg(sparsity_npoints=N_SP)
    model = models.NerfModel(pcfg)
    state = models.TrainState(pcfg, flat.clone())
    per = B_GLOBAL // world
    sl = slice(rank * per, (rank + 1) * per)
    calls = []

    def counted_all_reduce(t):
        calls.append((t.data_ptr(), t.numel()))
        comm.all_reduce_sum(t)

    reducer = dist.GradReducer(comm, all_reduce=counted_all_reduce)
    for step in range(STEPS):
        t_rand, u, sp = rnd[step]
        batch = {"rays": utils.Rays(*[r[sl].contiguous() for r in rays]), "pixels": px[sl].contiguous()}
        lr = utils.learning_rate_decay(step, 5e-4, 5e-6, 100)
        models.train_step(model, state, batch, lr, t_rand=t_rand[sl].contiguous(), u=u[sl].contiguous(),
                          sp_points=sp, world_size=comm.world, reducer=reducer)
    assert state.step == STEPS
    # lax.pmean(grad) + lax.pmean(stats) (train.py:117-118) = exactly two collectives per step, in this order: MLP_0's half
    # of the gradient arena (final after the coarse level: it rides under the fine level on the GPU), then MLP_1's half
    # with the 6 stats in its tail; together they cover the arena exactly once
    n_mlp = flat.numel() // 2
    base = state.reduce_buf.data_ptr()
    assert calls == [(base, n_mlp), (base + 4 * n_mlp, n_mlp + 8)] * STEPS, calls
    # render_image: padded chunks, per-rank slices, all-gather (nerf_sh/nerf/utils.py:357-371)
    H, W = 5, 7                                   # 35 rays: odd, chunk 16 -> padding on every chunk
    g = torch.Generator().manual_seed(3)
    img_rays = utils.Rays(*[torch.randn(H, W, 3, generator=g) for _ in range(3)])

    def render_fn(r):
        rgb = r.origins * 2 + r.directions
        return [(rgb, rgb[:, 0], rgb[:, 1]), (rgb + 1, rgb[:, 0] * 3, rgb[:, 2])]

    rgb, disp, acc = utils.render_image(render_fn, img_rays, chunk=16, world_size=comm.world, rank=comm.rank,
                                        gather=comm.all_gather_cat)
    # voxel-sharded grid evaluation + gather (octree/extraction.py step 1)
    sig = extraction.grid_sigma(model, state, 5, [0, 0, 0], [1.5, 1.5, 1.5], comm)       # ragged slabs (3 + 2 planes)
    sig6 = extraction.grid_sigma(model, state, 6, [0, 0, 0], [1.5, 1.5, 1.5], comm)      # equal slabs: gathered in place
    torch.save({"params": state.params, "stats": state.stats, "rgb": rgb, "disp": disp, "acc": acc, "sig": sig,
                "sig6": sig6},
               os.path.join(outdir, f"rank{rank}.pt"))
    comm.barrier()
    comm.shutdown()


@pytest.mark.timeout(600)
def test_two_rank_data_parallel_matches_single_process():
    world = 2
    with tempfile.TemporaryDirectory() as outdir:
        mp.spawn(_worker, args=(world, _free_port(), outdir), nprocs=world, join=True)
        res = [torch.load(os.path.join(outdir, f"rank{r}.pt")) for r in range(world)]
    # replicas stay bit-identical (same reduced gradient, same deterministic Adam)
    assert torch.equal(res[0]["params"], res[1]["params"])
    assert torch.equal(res[0]["stats"], res[1]["stats"])
    # single-process reference on the full batch with the same randoms
    cfg, flat, rays, px, rnd = _problem()
    m = torch.zeros_like(flat); v = torch.zeros_like(flat)
    for step in range(STEPS):
        t_rand, u, sp = rnd[step]
        lr = O.learning_rate_decay(step, 5e-4, 5e-6, 100)
        flat, m, v, stats, _ = O.train_step(flat, m, v, step, rays, px, cfg, t_rand, u, sp, lr)
    # After one Adam step every parameter moved by ~lr; the two runs may differ where a gradient is
    # ~0 (sign noise), so compare with an absolute tolerance of a fraction of lr.
    diff = (res[0]["params"] - flat).abs()
    assert float(diff.max()) < 2.5e-4 and float(diff.mean()) < 5e-6, (float(diff.max()), float(diff.mean()))
    np.testing.assert_allclose(res[0]["stats"][0].item(), float(stats["loss"]), rtol=1e-4)
    # render_image gather == single-process evaluation
    H, W = 5, 7
    g = torch.Generator().manual_seed(3)
    o, d, _ = [torch.randn(H, W, 3, generator=g) for _ in range(3)]
    rgb = o * 2 + d + 1
    for r in res:
        assert torch.allclose(r["rgb"], rgb) and torch.allclose(r["disp"][..., 0], (o * 2 + d)[..., 0] * 3)
        assert torch.allclose(r["acc"][..., 0], (o * 2 + d)[..., 2])
    # sharded grid == full grid, on every rank
    for reso, key in ((5, "sig"), (6, "sig6")):
        ix, iy, iz = torch.meshgrid(*[torch.arange(reso, dtype=torch.float32)] * 3, indexing="ij")
        full = (ix * 10000 + iy * 100 + iz).reshape(-1)
        for r in res:
            assert torch.equal(r[key], full)


def test_slab_range_partitions_grid():
    from plenoctree_amd import dist
    for reso, world in ((512, 8), (10, 4), (7, 8), (256, 1)):
        spans = [dist.slab_range(reso, world, r) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == reso
        for a, b in zip(spans[:-1], spans[1:]):
            assert a[1] == b[0]
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1


def _per_host_worker(rank, world, port, outdir):
    """`--per_host_image` (default auto = true for ranks sharing a host) on gloo ranks: the rank's dataset is a shard of host 0's draw; the shards, gathered in rank order
    with the collective render_image / the reducer use, must be the single-process batch."""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(1)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _cpu_feeder import feeder_for
    from plenoctree_amd import dist
    from plenoctree_amd.nerf_sh.nerf import datasets, utils
    datasets.Dataset.feeder_factory = staticmethod(feeder_for)
    comm = dist.init_from_env(backend="gloo")
    # the DEFAULT flags: `--per_host_image auto` resolves to the reference's single-host sampler because the ranks share a host
    args = utils.define_flags().parse_args(["--config", "synthetic", "--train_dir", "x"])
    utils.update_flags(args); args.factor = 16
    assert args.per_host_image == "auto" and dist.per_host_image(args.per_host_image, comm.world) is True
    assert dist.per_host_image("false", comm.world) is False and dist.per_host_image("auto", 1) is False
    os.environ["LOCAL_WORLD_SIZE"] = str(world // 2)          # a launcher that says: two hosts -> the multi-host sampler
    assert dist.per_host_image("auto", comm.world) is False and dist.per_host_image("true", comm.world) is True
    del os.environ["LOCAL_WORLD_SIZE"]
    per = 64 // world
    # (what nerf_sh/train.py builds for per_host_image = true)
    ds = datasets.get_dataset("train", args, torch.device("cpu"), batch_size=per, seed=20201473, shard=(comm.rank, comm.world))
    out = []
    for _ in range(3):
        b = next(ds)
        out.append({"pixels": comm.all_gather_cat(b["pixels"]), "rays": [comm.all_gather_cat(r) for r in b["rays"]]})
    if rank == 0:
        torch.save(out, os.path.join(outdir, "gathered.pt"))
    comm.barrier()
    comm.shutdown()


@pytest.mark.timeout(300)
def test_per_host_image_over_gloo_ranks_replays_the_single_process_batches():
    """The reference on one host with N devices draws ONE image and ONE set of batch_size pixel ids per step and shards them
    (nerf_sh/nerf/datasets.py:159-166, nerf_sh/nerf/utils.py:518-522): 4 gloo ranks x 16 rays, gathered in rank order, are the
    64-ray batches of a single process, step after step."""
    world = 4
    with tempfile.TemporaryDirectory() as outdir:
        mp.spawn(_per_host_worker, args=(world, _free_port(), outdir), nprocs=world, join=True)
        got = torch.load(os.path.join(outdir, "gathered.pt"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from plenoctree_amd.nerf_sh.nerf import datasets, utils
    args = utils.define_flags().parse_args(["--config", "synthetic", "--train_dir", "x"])
    utils.update_flags(args); args.factor = 16
    whole = datasets.get_dataset("train", args, torch.device("cpu"), batch_size=64, seed=20201473)
    for g in got:
        want = next(whole)
        assert torch.equal(g["pixels"], want["pixels"])
        for k in range(3):
            assert torch.equal(g["rays"][k], want["rays"][k])
