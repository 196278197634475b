"""The reference's ON-DISK data formats at the bit level (-m gpu): a scene exported by scripts/export_scene.py as a NeRF-Synthetic
directory (transforms_*.json + RGBA PNGs; nerf_sh/nerf/datasets.py:189-232) and as an NSVF directory (intrinsics.txt, bbox.txt,
pose/, rgb/; :491-552), read back through datasets.Blender / datasets.NSVF, must give the SAME cameras, images, batches, trained
parameters and eval PSNR as datasets.Synthetic with the same options (colours on the 8-bit grid a PNG holds).  A 4-view cut at
96 x 128 here; the full-size runs (100 views, 800 x 800 RGBA / 1920 x 1080) through the five CLIs are profiles/r06*_pipeline_cli_*.log.
"""
import os
import sys

import numpy as np
import pytest
import torch

from _helpers import _gpu, _ops

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H, W, VIEWS = 96, 128, (4, 2, 2)


def _args(dataset, data_dir=None, sh_deg=3):
    from plenoctree_amd.nerf_sh.nerf import utils
    args = utils.define_flags().parse_args(["--config", "blender" if dataset != "nsvf" else "tt", "--train_dir", "x"])
    utils.update_flags(args)
    args.dataset, args.data_dir, args.factor, args.batch_size = dataset, data_dir, 0, 256
    args.near, args.far, args.sparsity_radius, args.sparsity_length, args.sh_deg = 2.0, 6.0, 1.5, 0.05, sh_deg
    args.sparsity_npoints = 500
    args.synthetic_hw, args.synthetic_views, args.synthetic_8bit = (H, W), (VIEWS[0], max(VIEWS[1:])), True
    return args


@pytest.mark.parametrize("fmt", ["blender", "nsvf"])
def test_on_disk_format_replays_the_synthetic_scene(tmp_path, fmt):
    _ops(); dev = _gpu()
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import export_scene
    from plenoctree_amd.nerf_sh.nerf import datasets, models, utils
    path = str(tmp_path / fmt)
    (export_scene.write_blender if fmt == "blender" else export_scene.write_nsvf)(path, H, W, *VIEWS, dev)
    a_disk, a_syn = _args(fmt, path), _args("synthetic")
    if fmt == "blender":
        assert sorted(os.listdir(path)) == ["test", "train", "transforms_test.json", "transforms_train.json", "transforms_val.json", "val"]
        from PIL import Image
        assert Image.open(os.path.join(path, "train", "r_0.png")).mode == "RGBA"
    else:
        assert sorted(os.listdir(path)) == ["bbox.txt", "intrinsics.txt", "pose", "rgb"]
    disk = datasets.get_dataset("train", a_disk, dev)
    syn = datasets.get_dataset("train", a_syn, dev)
    assert (disk.h, disk.w, disk.n_examples) == (H, W, VIEWS[0]) == (syn.h, syn.w, syn.n_examples)
    assert disk.focal == syn.focal
    np.testing.assert_array_equal(disk.camtoworlds, syn.camtoworlds)
    assert torch.equal(disk.images, syn.images)                       # the decoded (and composited) PNGs == the 8-bit scene
    assert float(disk.images.min()) < 0.5 and int((disk.images != 1.0).sum()) > 1000      # ... and there is something in them
    if fmt == "nsvf":
        np.testing.assert_allclose(disk.bbox, [-1.2, -1.2, -1.2, 1.2, 1.2, 1.2])
    # the same batches, step after step (one image per step, 256 random pixels: datasets.py:159-166)
    for _ in range(3):
        b0, b1 = next(disk), next(syn)
        assert torch.equal(b0["pixels"], b1["pixels"])
        for r0, r1 in zip(b0["rays"], b1["rays"]):
            assert torch.equal(r0, r1)
    # 40 train steps from the fixed-seed initialisation: bit-identical parameters, hence the same eval PSNR on a held-out view
    out = []
    for args, ds_cls in ((a_disk, None), (a_syn, None)):
        model, params = models.construct_nerf(args, dev)
        state = models.TrainState(model.cfg, params)
        ds = datasets.get_dataset("train", args, dev)
        for step in range(40):
            lr = utils.learning_rate_decay(step, args.lr_init, args.lr_final, args.max_steps)
            models.train_step(model, state, next(ds), lr, randomized=True, seed=step << 8)
        test = datasets.get_dataset("test", args, dev)
        ex = test.get_image(1)
        rgb, _, _ = utils.render_image(lambda r: model.apply(state, r, False), ex["rays"], chunk=4096)
        out.append((state.params.clone(), utils.compute_psnr(((rgb - ex["pixels"]) ** 2).mean().item()), ex["pixels"]))
    assert torch.equal(out[0][2], out[1][2])                          # the held-out view itself
    assert torch.equal(out[0][0], out[1][0]), "training on the exported files diverged from training on the scene"
    assert abs(out[0][1] - out[1][1]) <= 1e-3 and out[0][1] > 8.0, (out[0][1], out[1][1])
