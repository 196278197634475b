"""N > 1 host logic of the PlenOctree side on CPU: two gloo ranks run the product's extraction.step2,
extraction.calculate_grid_weights, extraction.eval_octree and optimization.fit with the HIP entry points
monkeypatched by oracle-backed stand-ins (test infrastructure only -- the product path has no CPU fallback),
and the results are compared with a single-process restatement."""
import os
import socket
import sys
import tempfile
import types

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import octree_oracle as T  # noqa: E402

K, DEPTH, W, H, FX = 4, 2, 5, 4, 4.5
N_TRAIN, N_VAL, LR, EPOCHS = 3, 3, 0.5, 2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _poses(n, seed):
    from plenoctree_amd.nerf_sh.nerf.datasets import pose_spherical
    rs = np.random.RandomState(seed)
    return np.stack([pose_spherical(rs.uniform(0, 360), rs.uniform(-10, 60), 4.0) for _ in range(n)])


def _oracle_tree():
    mask = np.random.RandomState(0).rand(8, 8, 8) < 0.3
    t = T.build_from_mask(mask, DEPTH, 3 * K + 1, [0.0, 0.1, 0.0], [1.3, 1.2, 1.4])
    rs = np.random.RandomState(1)
    t.data[:] = (rs.randn(*t.data.shape) * 0.5).astype(np.float32)
    t.data[..., -1] = (np.abs(rs.randn(*t.data.shape[:-1])) * 3).astype(np.float32)
    return t


def _host_tree(t):
    from plenoctree_amd.octree.svox import N3Tree
    radius = 0.5 / t.invradius
    h = N3Tree(N=2, data_dim=t.data_dim, depth_limit=DEPTH, radius=radius, center=(1.0 - 2.0 * t.offset) * radius,
               data_format=f"SH{K}")
    h.child, h.parent_depth = torch.from_numpy(t.child.copy()), torch.from_numpy(t.parent_depth.copy())
    h.data = torch.from_numpy(t.data.copy())
    h.level_nodes = np.bincount(t.parent_depth[:, 1]).tolist()
    return h


def _rays(c2w):
    rays = [T.cam2world_ray(ix, iy, np.asarray(c2w), W, H, FX, FX) for iy in range(H) for ix in range(W)]
    return np.stack([r[0] for r in rays]), np.stack([r[1] for r in rays])


def _render64(t, data, c2w, opt):
    ro, rd = _rays(c2w)
    return T.render_rays_torch(t, data, ro, rd, rd, opt).reshape(H, W, 3)


def _patch(structure):
    """Oracle-backed stand-ins for the octree_ops / ops entry points the drivers call."""
    from plenoctree_amd import octree_ops as oops, ops

    def tree_view(child, data, offset, invradius):
        return types.SimpleNamespace(data=data)

    def _opt(o):
        return T.RenderOptions(o.step_size, o.background_brightness, o.sigma_thresh, o.stop_thresh)

    def octree_render_persp(view, c2w, width, height, fx, opts, fy=None):
        return _render64(structure, view.data.double(), c2w.numpy(), _opt(opts)).float()

    def octree_render_persp_bwd(view, c2w, width, height, fx, opts, grad_out, grad_data, fy=None, out_rgb=None):
        with torch.enable_grad():              # the drivers run under torch.no_grad(); the stand-in differentiates the oracle
            d = view.data.double().detach().requires_grad_(True)
            (_render64(structure, d, c2w.numpy(), _opt(opts)) * grad_out.double()).sum().backward()
        grad_data += d.grad.float()
        return grad_data

    def image_mse(im, gt, want_grad=True):
        with torch.enable_grad():
            x = im.detach().clone().requires_grad_(True)
            sse = ((x.clamp(0.0, 1.0) - gt) ** 2).sum()
            if want_grad:
                (sse / x.numel()).backward()
        return sse.detach().reshape(1), (x.grad if want_grad else None)

    def sgd_step(params, grads, lr, momentum=0.0, nesterov=False, buf=None, first_step=False):
        params -= lr * grads

    def grid_weight_render(sigma_grid, reso, c2w_all, fx, fy, width, height, opts, offset, invradius, grid_weight=None):
        # any camera-dependent, order-independent per-voxel quantity exercises the sharding + max-reduce
        for c in c2w_all:
            grid_weight.copy_(torch.maximum(grid_weight, torch.sin(sigma_grid * float(c[:3, 3].sum())).abs()))
        return grid_weight

    def tree_sample_cells(parent_depth, node0, n_nodes, samples, offset, invradius, u=None, seed=0, stream_id=0):
        n = torch.arange(node0, node0 + n_nodes, dtype=torch.float32)
        cell = torch.arange(8, dtype=torch.float32)
        s = torch.arange(samples, dtype=torch.float32)
        base = (n[:, None, None] * 8 + cell[None, :, None]) + 0.01 * s[None, None, :]
        return torch.stack([base, base * 0.5, base * 0.25], -1).reshape(n_nodes * 8, samples, 3)

    def mean_over_samples(cfg, raw_rgb, raw_sigma, samples_per_cell, out=None):
        val = torch.cat([raw_rgb, raw_sigma], -1).reshape(-1, samples_per_cell, raw_rgb.shape[-1] + 1).mean(1)
        out.copy_(val)
        return out

    oops.tree_view, oops.octree_render_persp, oops.octree_render_persp_bwd = tree_view, octree_render_persp, octree_render_persp_bwd
    oops.image_mse, oops.sgd_step, oops.grid_weight_render, oops.tree_sample_cells = image_mse, sgd_step, grid_weight_render, tree_sample_cells
    oops.render_opts = lambda step_size=1e-3, background_brightness=1.0, sigma_thresh=0.0, stop_thresh=0.0: types.SimpleNamespace(
        step_size=step_size, background_brightness=background_brightness, sigma_thresh=sigma_thresh, stop_thresh=stop_thresh)
    ops.mean_over_samples = mean_over_samples


class _Model:
    cfg = None

    @staticmethod
    def eval_points_raw(state, pts):
        rgb = torch.stack([pts[:, 0] * (c + 1) + pts[:, 1] for c in range(3 * K)], -1)
        return rgb, (pts[:, 2:3] - 3.0)


def _problem():
    t = _oracle_tree()
    train_c2w, val_c2w = _poses(N_TRAIN, 1), _poses(N_VAL, 2)
    rs = np.random.RandomState(3)
    train_gt = [torch.from_numpy(rs.rand(H, W, 3).astype(np.float32)) for _ in range(N_TRAIN)]
    val_gt = [torch.from_numpy(rs.rand(H, W, 3).astype(np.float32)) for _ in range(N_VAL)]
    args = types.SimpleNamespace(dp_grad_reduce="sum", renderer_step_size=1e-2, sgd=True, lr=LR, sgd_momentum=0.0, sgd_nesterov=False,
                                 num_epochs=EPOCHS, val_interval=1, continue_on_decrease=True, no_early_stop=True,
                                 samples_per_cell=3, chunk=48)
    return t, args, (torch.from_numpy(train_c2w), train_gt), (torch.from_numpy(val_c2w), val_gt)


def _worker(rank, world, port, outdir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from plenoctree_amd import dist
    from plenoctree_amd.octree import extraction, optimization
    comm = dist.init_from_env(backend="gloo")
    t, args, train, val = _problem()
    _patch(t)
    tree = _host_tree(t)
    # fine-tuning: one gradient per rank and image group, summed by all-reduce
    history, best = optimization.fit(args, tree, train, val, H, W, FX, comm, say=lambda *a, **k: None)
    # validation set as a dataset object for eval_octree
    ds = types.SimpleNamespace(size=N_VAL, w=W, h=H, focal=FX, camtoworlds=val[0].numpy(),
                               get_image=lambda i: {"pixels": val[1][i]})
    psnr, _ = extraction.eval_octree(tree, ds, args, comm)
    # step 2 on a fresh copy of the structure: nodes sharded, blocks all-gathered
    tree2 = _host_tree(t)
    tree2.data.data.zero_()
    extraction.step2(args, tree2, _Model, None, comm, seed=0)
    # weight mask: cameras sharded, max-reduced
    sig = torch.linspace(-2.0, 2.0, 8 ** 3)
    cams = types.SimpleNamespace(camtoworlds=_poses(5, 9), focal=FX, w=W, h=H)
    wts = extraction.calculate_grid_weights(cams, sig, 8, tree.invradius, tree.offset, 1e-2, comm)
    torch.save({"data": tree.data, "history": history, "psnr": psnr, "step2": tree2.data, "weights": wts},
               os.path.join(outdir, f"rank{rank}.pt"))
    comm.barrier()
    comm.shutdown()


@pytest.mark.timeout(600)
def test_two_rank_octree_drivers_match_single_process():
    world = 2
    with tempfile.TemporaryDirectory() as outdir:
        mp.spawn(_worker, args=(world, _free_port(), outdir), nprocs=world, join=True)
        res = [torch.load(os.path.join(outdir, f"rank{r}.pt"), weights_only=False) for r in range(world)]
    assert torch.equal(res[0]["data"], res[1]["data"])               # replicas stay identical
    assert res[0]["history"] == res[1]["history"] and res[0]["psnr"] == res[1]["psnr"]
    # independent restatement: summed gradient of each group of `world` images (--dp_grad_reduce sum), plain SGD
    t, args, (train_c2w, train_gt), (val_c2w, val_gt) = _problem()
    opt = T.RenderOptions(args.renderer_step_size)
    data = torch.from_numpy(t.data.copy()).double()

    def grad_of(j):
        d = data.clone().requires_grad_(True)
        im = _render64(t, d, train_c2w[j].numpy(), opt).float()
        ((im.clamp(0.0, 1.0) - train_gt[j]) ** 2).mean().backward()
        return d.grad

    def val_psnr():
        ps = []
        for j in range(N_VAL):
            im = _render64(t, data, val_c2w[j].numpy(), opt).float()
            ps.append(-10.0 * np.log10(float(((im.clamp(0, 1) - val_gt[j]) ** 2).mean())))
        return float(np.mean(ps))

    want_hist = [val_psnr()]
    for _ in range(EPOCHS):
        for j0 in range(0, N_TRAIN, world):
            group = list(range(j0, min(j0 + world, N_TRAIN)))
            g = sum(grad_of(j) for j in group)
            data = (data.float() - LR * g.float()).double()
        want_hist.append(val_psnr())
    got_hist = [h[2] for h in res[0]["history"]]
    np.testing.assert_allclose(got_hist, want_hist, rtol=0, atol=2e-4)
    assert float((res[0]["data"].double() - data).abs().max()) < 2e-5
    np.testing.assert_allclose(res[0]["psnr"], want_hist[-1], atol=2e-4)
    # step 2: every deepest-level cell holds the mean of the stand-in network at its own samples
    levels = np.bincount(t.parent_depth[:, 1])
    node0 = t.n_internal - levels[-1]
    for r in res:
        got = r["step2"][node0:].reshape(-1, 3 * K + 1)
        base = torch.arange(node0 * 8, t.n_internal * 8, dtype=torch.float32)[:, None] + 0.01 * torch.arange(3.0)[None]
        rgb = torch.stack([base * (c + 1) + base * 0.5 for c in range(3 * K)], -1).mean(1)
        np.testing.assert_allclose(got[:, :-1], rgb, rtol=1e-6)
        np.testing.assert_allclose(got[:, -1], (base * 0.25 - 3.0).mean(1), rtol=1e-6)
        assert not bool(r["step2"][:node0].any())
    # weight mask: maximum over all five cameras
    sig = torch.linspace(-2.0, 2.0, 8 ** 3)
    want = torch.zeros_like(sig)
    for c in _poses(5, 9):
        want = torch.maximum(want, torch.sin(sig * float(c[:3, 3].sum())).abs())
    for r in res:
        assert torch.equal(r["weights"], want)
