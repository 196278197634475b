"""Host-side logic of the reference-mirroring layer (flags, presets, schedules, feeders,
checkpoints) -- runs without a GPU."""
import os

import numpy as np
import pytest
import torch

from oracle import nerf_oracle as O
from plenoctree_amd.nerf_sh.nerf import datasets, utils


def _args(argv=()):
    return utils.define_flags().parse_args(list(argv))


def test_presets_and_flag_errors(tmp_path):
    a = _args(["--config", "blender", "--train_dir", "x"])
    utils.update_flags(a)
    assert (a.sh_deg, a.num_coarse_samples, a.num_fine_samples, a.use_viewdirs, a.batch_size) == (3, 64, 128, False, 1024)
    a = _args(["--config", "tt", "--train_dir", "x"])
    utils.update_flags(a)
    assert (a.sh_deg, a.near, a.far, a.sparsity_radius, a.sparsity_length) == (4, 0.0, 4.0, 5.0, 0.2)
    bad = tmp_path / "bad.yaml"
    bad.write_text("not_a_flag: 1\n")
    a = _args(["--config", str(bad)])
    with pytest.raises(ValueError):            # nerf_sh/nerf/utils.py:241-243
        utils.update_flags(a)
    with pytest.raises(ValueError):            # train_dir must be set (:248-249)
        utils.check_flags(_args([]))
    a = _args(["--train_dir", "x", "--data_dir", "y"])   # defaults: use_viewdirs=True, sh_deg=-1
    with pytest.raises(NotImplementedError):
        utils.check_flags(a)
    a = _args(["--config", "blender", "--train_dir", "x", "--data_dir", "y", "--batch_size", "1000"])
    utils.update_flags(a); a.batch_size = 1001
    with pytest.raises(ValueError):            # batch divisible by device count (:252)
        utils.check_flags(a, require_batch_size_div=True, world_size=8)


def test_lr_psnr_rays_match_oracle():
    for step in (0, 1, 500, 999999, 2000000, 3000000):
        assert utils.learning_rate_decay(step, 5e-4, 5e-6, 2000000) == pytest.approx(
            O.learning_rate_decay(step, 5e-4, 5e-6, 2000000), rel=1e-12)
    assert utils.learning_rate_decay(10, 5e-4, 5e-6, 1000, 100, 0.01) == pytest.approx(
        O.learning_rate_decay(10, 5e-4, 5e-6, 1000, 100, 0.01), rel=1e-12)
    assert utils.compute_psnr(1e-3) == pytest.approx(30.0)
    c2w = np.stack([datasets.pose_spherical(30.0, 20.0, 4.0311), datasets.pose_spherical(200.0, -5.0, 4.0311)])
    r1, r2 = utils.generate_rays(9, 7, 12.0, c2w), O.generate_rays(9, 7, 12.0, c2w)
    for a, b in zip(r1, r2):
        np.testing.assert_array_equal(a, b)
    # cameras look at the origin along -z
    for m in c2w:
        fwd = -m[:3, 2]
        np.testing.assert_allclose(fwd, -m[:3, 3] / np.linalg.norm(m[:3, 3]), atol=1e-6)


def test_synthetic_dataset_batches():
    a = _args(["--config", "blender", "--train_dir", "x", "--dataset", "synthetic"])
    utils.update_flags(a); a.dataset = "synthetic"; a.factor = 8
    ds = datasets.get_dataset("train", a, torch.device("cpu"), batch_size=64)
    assert ds.h == ds.w == 100 and ds.size == 100
    b = next(ds)
    assert b["pixels"].shape == (64, 3) and b["rays"].origins.shape == (64, 3)
    assert float(b["pixels"].min()) >= 0 and float(b["pixels"].max()) <= 1
    np.testing.assert_allclose(b["rays"].viewdirs.norm(dim=-1).numpy(), 1.0, rtol=1e-5)
    # per-pixel rays agree with generate_rays (nerf_sh/nerf/utils.py:545-589)
    img = datasets.get_dataset("test", a, torch.device("cpu")).get_image(3)
    ds_t = datasets.get_dataset("test", a, torch.device("cpu"))
    ref = utils.generate_rays(ds_t.w, ds_t.h, ds_t.focal, ds_t.camtoworlds[3:4])
    np.testing.assert_allclose(img["rays"].directions.numpy(), ref.directions[0], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(img["rays"].origins.numpy(), ref.origins[0], rtol=1e-6)
    # same seed -> same batch sequence (np.random.seed(20201473 + host_id), train.py:128)
    d1 = datasets.get_dataset("train", a, torch.device("cpu"), batch_size=8)
    d2 = datasets.get_dataset("train", a, torch.device("cpu"), batch_size=8)
    assert torch.equal(next(d1)["pixels"], next(d2)["pixels"])
    with pytest.raises(NotImplementedError):
        a.dataset = "llff"
        datasets.get_dataset("train", a, torch.device("cpu"))


def test_per_host_image_shards_replay_the_single_device_batch():
    """--per_host_image (datasets shard=(rank, world)): the reference on ONE host draws one image and batch_size pixel ids per
    step and shards them over its local devices (nerf_sh/nerf/datasets.py:159-166 + utils.shard, utils.py:518-522).  8 ranks of
    512 rays must replay the 1 x 4096 batches exactly, step after step; without the option every rank is its own host and
    draws its own image (train.py:128)."""
    a = _args(["--config", "blender", "--train_dir", "x", "--dataset", "synthetic"])
    utils.update_flags(a); a.dataset = "synthetic"; a.factor = 8
    cpu = torch.device("cpu")
    whole = datasets.get_dataset("train", a, cpu, batch_size=4096, seed=20201473)
    shards = [datasets.get_dataset("train", a, cpu, batch_size=512, seed=20201473, shard=(r, 8)) for r in range(8)]
    for _ in range(3):
        want = next(whole)
        got = [next(d) for d in shards]
        assert torch.equal(torch.cat([g["pixels"] for g in got]), want["pixels"])
        for k in range(3):
            assert torch.equal(torch.cat([g["rays"][k] for g in got]), want["rays"][k])
    # the reference's pieces are utils.shard's: [n_devices, B / n_devices] in row order
    assert torch.equal(utils.shard(want["pixels"], 8, 5), got[5]["pixels"])
    # default sampler: rank r is host r
    own = [datasets.get_dataset("train", a, cpu, batch_size=512, seed=20201473 + r) for r in range(2)]
    b0, b1 = next(own[0]), next(own[1])
    assert not torch.equal(b0["rays"].origins, b1["rays"].origins)            # different images
    with pytest.raises(ValueError):
        datasets.get_dataset("train", a, cpu, batch_size=512, shard=(8, 8))
    a.image_batching = True
    with pytest.raises(ValueError):
        next(datasets.get_dataset("train", a, cpu, batch_size=512, shard=(0, 8)))


def test_blender_loader(tmp_path):
    from PIL import Image
    import json
    frames = []
    os.makedirs(tmp_path / "train")
    for i in range(3):
        rgba = (np.random.RandomState(i).rand(8, 8, 4) * 255).astype(np.uint8)
        Image.fromarray(rgba, "RGBA").save(tmp_path / "train" / f"r_{i}.png")
        frames.append({"file_path": f"./train/r_{i}", "transform_matrix": datasets.pose_spherical(40 * i, 30, 4).tolist()})
    (tmp_path / "transforms_train.json").write_text(json.dumps({"camera_angle_x": 0.6911112, "frames": frames}))
    a = _args(["--config", "blender", "--train_dir", "x", "--data_dir", str(tmp_path)])
    utils.update_flags(a)
    ds = datasets.get_dataset("train", a, torch.device("cpu"), batch_size=16)
    assert (ds.h, ds.w, ds.size) == (8, 8, 3)
    assert ds.focal == pytest.approx(0.5 * 8 / np.tan(0.5 * 0.6911112))
    rgba = np.asarray(Image.open(tmp_path / "train" / "r_1.png"), np.float32) / 255
    want = rgba[..., :3] * rgba[..., 3:] + (1 - rgba[..., 3:])           # white background composite
    np.testing.assert_allclose(ds.images[1].numpy().reshape(8, 8, 3), want, rtol=1e-6)
    b = next(ds)
    assert b["pixels"].shape == (16, 3)
    # image_batching (datasets.py:137-141,152-157): rays drawn from the flattened table of ALL images; every
    # drawn (ray, pixel) pair must be the pair of its own image
    a.image_batching = True
    ds_all = datasets.get_dataset("train", a, torch.device("cpu"), batch_size=64)
    b = next(ds_all)
    full = [ds_all.get_image(i) for i in range(3)]
    table_o = torch.cat([f["rays"].origins.reshape(-1, 3) for f in full])
    table_d = torch.cat([f["rays"].directions.reshape(-1, 3) for f in full])
    table_px = torch.cat([f["pixels"].reshape(-1, 3) for f in full])
    seen = set()
    for i in range(64):
        hit = ((table_d - b["rays"].directions[i]).abs().sum(-1) < 1e-6) & ((table_o - b["rays"].origins[i]).abs().sum(-1) < 1e-6)
        assert int(hit.sum()) == 1
        j = int(hit.nonzero()[0])
        seen.add(j // 64)
        assert torch.equal(table_px[j], b["pixels"][i])
    assert len(seen) > 1                       # one batch spans several images


def test_nsvf_loader_on_a_synthetic_directory(tmp_path):
    """NSVF-format directory (reference nerf_sh/nerf/datasets.py:491-552): split prefixes, OpenCV->OpenGL pose
    flip, RGBA on white, focal from intrinsics, bbox.txt, test split falling back to 1_."""
    from PIL import Image
    root = tmp_path / "scene"
    (root / "pose").mkdir(parents=True); (root / "rgb").mkdir()
    K = np.eye(4); K[0, 0], K[1, 1], K[0, 2], K[1, 2] = 30.0, 32.0, 4.0, 3.0
    np.savetxt(root / "intrinsics.txt", K)
    np.savetxt(root / "bbox.txt", np.array([[-1.0, -2.0, -3.0, 1.0, 2.0, 3.0, 0.4]]))
    rs = np.random.RandomState(0)
    poses = {}
    for name in ("0_0000", "0_0001", "0_0002", "1_0000"):
        pose = np.eye(4); pose[:3, :3] = np.linalg.qr(rs.randn(3, 3))[0]; pose[:3, 3] = rs.randn(3)
        poses[name] = pose
        np.savetxt(root / "pose" / f"{name}.txt", pose)
        rgba = (rs.rand(6, 8, 4) * 255).astype(np.uint8)
        Image.fromarray(rgba, "RGBA").save(root / "rgb" / f"{name}.png")
        poses[name + "_img"] = rgba
    a = _args(["--config", "tt", "--train_dir", "x", "--data_dir", str(root)])
    utils.update_flags(a); a.factor = 0
    ds = datasets.get_dataset("train", a, torch.device("cpu"), batch_size=16)
    assert (ds.size, ds.h, ds.w) == (3, 6, 8) and ds.focal == pytest.approx(31.0)
    np.testing.assert_allclose(ds.bbox, [-1, -2, -3, 1, 2, 3])
    np.testing.assert_allclose(ds.camtoworlds[1], poses["0_0001"] @ np.diag([1.0, -1.0, -1.0, 1.0]), rtol=1e-6)
    rgba = poses["0_0002_img"].astype(np.float32) / 255.0
    want = rgba[..., :3] * rgba[..., 3:] + (1.0 - rgba[..., 3:])
    np.testing.assert_allclose(ds.get_image(2)["pixels"].numpy(), want, atol=1e-6)
    batch = next(ds)
    assert batch["pixels"].shape == (16, 3) and batch["rays"].origins.shape == (16, 3)
    test = datasets.get_dataset("test", a, torch.device("cpu"))
    assert test.size == 1                                        # no 2_ files: falls back to 1_
    np.testing.assert_allclose(test.camtoworlds[0], poses["1_0000"] @ np.diag([1.0, -1.0, -1.0, 1.0]), rtol=1e-6)
    a.factor = 2
    half = datasets.get_dataset("train", a, torch.device("cpu"), batch_size=4)
    assert (half.h, half.w) == (3, 4) and half.focal == pytest.approx(15.5)


def test_area_resize_full_size_is_block_mean_and_fast():
    """The cv2.INTER_AREA stand-in at the reference's real size (800 x 800 RGBA -> 400 x 400, datasets.py:208-212): equal to
    the 2 x 2 block mean, a non-integer ratio keeps the image mean, and an image takes well under a second (a naive
    three-operand einsum took minutes)."""
    import time
    from plenoctree_amd.nerf_sh.nerf.datasets import area_resize
    rs = np.random.RandomState(3)
    img = rs.rand(800, 800, 4).astype(np.float32)
    t0 = time.perf_counter()
    half = area_resize(img, 400, 400)
    dt = time.perf_counter() - t0
    want = img.reshape(400, 2, 400, 2, 4).astype(np.float64).mean(axis=(1, 3))
    np.testing.assert_allclose(half, want, atol=1e-6)
    assert dt < 5.0, f"area_resize of one 800 x 800 image took {dt:.1f} s"
    odd = area_resize(img, 300, 500)                                  # 800/300 is not an integer
    assert odd.shape == (300, 500, 4) and abs(float(odd.mean()) - float(img.mean())) < 1e-5


def test_noise_std_flag_reaches_the_cfg():
    """--noise_std (nerf_sh/nerf/utils.py:137-140; None in every preset) is accepted and becomes PxoCfg.noise_std; None = 0."""
    from plenoctree_amd.nerf_sh.nerf import models
    a = _args(["--config", "blender", "--train_dir", "x", "--noise_std", "0.25"])
    utils.update_flags(a)
    utils.check_supported(a)
    assert models.make_cfg(a).noise_std == pytest.approx(0.25)
    a.noise_std = None
    assert models.make_cfg(a).noise_std == 0.0
    a.noise_std = -1.0
    with pytest.raises(NotImplementedError):
        utils.check_supported(a)


def test_bench_self_launch_command():
    """`python bench.py --gpus N` without a launcher re-executes itself through torch.distributed.run, one rank per GPU
    (rendezvous on 127.0.0.1); with too few devices it says so instead of asking for a wrapper."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    cmd = bench.launch_command(4, ["--gpus", "4", "--steps", "7"], 23456)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "23456"
    assert cmd[-5:] == [os.path.join(root, "bench.py"), "--gpus", "4", "--steps", "7"]
    if not torch.cuda.is_available():
        env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], capture_output=True, text=True, env=env)
        assert r.returncode != 0 and "only 0 ROCm device(s) visible" in (r.stderr + r.stdout)


def test_every_reference_flag_parses():
    """A command line written for the reference must parse here: every flag the reference's entry points define
    (absl `flags.DEFINE_*` in nerf_sh/nerf/utils.py, octree/nerf/utils.py and the drivers) exists in the matching parser.
    The list is the reference's own, frozen here (the reference tree is not available on the GPU box)."""
    from plenoctree_amd.nerf_sh import train, eval as eval_mod, gen_video, gen_mesh
    from plenoctree_amd.nerf_sh.nerf import utils
    from plenoctree_amd.octree import extraction, optimization, evaluation
    common = ["train_dir", "data_dir", "config", "dataset", "image_batching", "white_bkgd", "batch_size", "factor", "spherify",
              "render_path", "llffhold", "model", "near", "far", "net_depth", "net_width", "net_depth_condition",
              "net_width_condition", "weight_decay_mult", "skip_layer", "num_rgb_channels", "num_sigma_channels", "randomized",
              "min_deg_point", "max_deg_point", "deg_view", "num_coarse_samples", "num_fine_samples", "use_viewdirs", "sh_deg",
              "sg_dim", "noise_std", "lindisp", "net_activation", "rgb_activation", "sigma_activation", "legacy_posenc_order",
              "lr_init", "lr_final", "lr_delay_steps", "lr_delay_mult", "max_steps", "save_every", "print_every",
              "render_every", "gc_every", "sparsity_weight", "sparsity_length", "sparsity_radius", "sparsity_npoints",
              "eval_once", "save_output", "chunk", "approx_eval_skip"]
    per_driver = {
        extraction: ["output", "center", "radius", "alpha_thresh", "max_refine_prop", "z_min", "z_max", "tree_branch_n",
                     "init_grid_depth", "samples_per_cell", "is_jaxnerf_ckpt", "masking_mode", "weight_thresh",
                     "projection_samples", "bbox_from_data", "data_bbox_scale", "autoscale", "bbox_cube", "bbox_scale",
                     "scale_alpha_thresh", "eval", "renderer_step_size", "no_early_stop"],
        optimization: ["input", "output", "render_interval", "val_interval", "num_epochs", "sgd", "lr", "sgd_momentum",
                       "sgd_nesterov", "write_vid", "split_train", "split_holdout_prop", "nosave", "continue_on_decrease",
                       "renderer_step_size", "no_early_stop"],
        evaluation: ["input", "write_vid", "write_images", "renderer_step_size", "no_early_stop"],
        gen_video: ["elevation", "num_views", "height", "width", "camera_angle_x", "intrin", "radius", "fps", "up_axis",
                    "write_poses"],
        gen_mesh: ["reso", "c1", "c2", "iso", "coarse", "point_chunk"],
    }
    def dests(parser):
        return {a.dest for a in parser._actions}
    base = dests(utils.define_flags())
    missing = [f for f in common if f not in base]
    assert not missing, missing
    for mod, names in per_driver.items():
        have = dests(mod.define_flags())
        missing = [f for f in names + common if f not in have]
        assert not missing, (mod.__name__, missing)
    assert hasattr(train, "main") and hasattr(eval_mod, "main")


def test_eval_output_files(tmp_path):
    """nerf_sh/eval.py:107-129: file names and contents of what the evaluation leaves in <train_dir>/test_preds."""
    from PIL import Image
    from plenoctree_amd.nerf_sh import eval as eval_mod
    rgb = torch.rand(5, 7, 3) * 1.4 - 0.2                      # values outside [0, 1] are clipped (utils.py:469-480)
    disp = torch.rand(5, 7)
    eval_mod.save_outputs(str(tmp_path), 3, rgb, disp)
    im = np.asarray(Image.open(tmp_path / "003.png"))
    assert im.shape == (5, 7, 3) and np.array_equal(im, (np.clip(rgb.numpy(), 0.0, 1.0) * 255.0).astype(np.uint8))
    dm = np.asarray(Image.open(tmp_path / "disp_003.png"))
    assert dm.shape == (5, 7) and np.array_equal(dm, (disp.numpy() * 255.0).astype(np.uint8))
    eval_mod.save_summary(str(tmp_path), 1234, [20.0, 30.0], [0.5, 0.75])
    assert float(open(tmp_path / "psnr.txt").read()) == 25.0 and open(tmp_path / "ssims_1234.txt").read() == "0.5 0.75"


def test_tile_schedule_properties(tmp_path):
    """The persistent-tile schedule of the fused MLP kernels (pxo_common.h: tile_sched - whole rounds of 128-row tiles, then
    at most one round of 64-row half tiles) covers [0, M) exactly, never uses more than one half round, and never needs more
    relu-mask / bias-partial slots than `mask_slots` provides: checked on the REAL header, compiled as host code
    (tests/native/sched_check.cpp), for every M up to 5000, around every half-round boundary and at the BASELINE sizes, for
    11 grid sizes.  A wrong schedule would silently drop or double rows for particular batch sizes only."""
    import shutil
    import subprocess
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not (os.path.exists(hipcc) or shutil.which(hipcc)):
        pytest.skip("needs hipcc")
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "sched_check")
    cmd = [hipcc, "-O1", "-std=c++17", "-x", "c++", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
           "-I" + os.path.join(ROOT, "plenoctree_amd", "csrc"), "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "native", "sched_check.cpp"), "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True, timeout=300)
    res = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0 and " bad 0 " in res.stdout, res.stdout[-2000:]
    assert int(res.stdout.split("cases ")[1].split()[0]) > 100000


def test_device_helpers_on_the_host(tmp_path, golden_dir):
    """The kernels' own helper functions, compiled as host code and run on the CPU (tests/native/device_helpers_check.cpp):
    `sh_basis<0..4>` of pxo_sh.h - the code the shading kernels and the octree renderer execute - against the reference's
    `octree/nerf/sh_proj.py:EvalSH` vectors (tests/golden/sh_proj.npz), and the parameter-arena layout of pxo_common.h against
    the reference's layer shapes and parameter counts."""
    import shutil
    import subprocess
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not (os.path.exists(hipcc) or shutil.which(hipcc)):
        pytest.skip("needs hipcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    g = np.load(os.path.join(golden_dir, "sh_proj.npz"))
    raw = str(tmp_path / "sh.bin")
    np.concatenate([g["dirs"].astype(np.float64).reshape(-1), g["basis"].astype(np.float64).reshape(-1)]).tofile(raw)
    exe = str(tmp_path / "device_helpers_check")
    cmd = [hipcc, "-O1", "-std=c++17", "-x", "c++", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
           "-I" + os.path.join(root, "plenoctree_amd", "csrc"), "-I" + os.path.join(root, "include"),
           os.path.join(root, "tests", "native", "device_helpers_check.cpp"), "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True, timeout=300)
    res = subprocess.run([exe, raw], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "dirs 64 " in res.stdout and "layout_bad 0" in res.stdout


def test_philox_emulator_against_random123_known_answers():
    """The Philox4x32-10 restatement the GPU generators are compared with (tests/_helpers.py) reproduces the published
    Random123 known-answer vectors (kat_vectors: philox4x32 10)."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from _helpers import philox4x32_10
    assert philox4x32_10((0, 0, 0, 0), (0, 0)) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert philox4x32_10((0xFFFFFFFF,) * 4, (0xFFFFFFFF,) * 2) == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert philox4x32_10((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0)) == \
        [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]
