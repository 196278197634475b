"""flax-msgpack checkpoint interop, CPU only.

The files written by plenoctree_amd.nerf_sh.nerf.checkpoints are fed to the REFERENCE's own
consumer, octree/nerf/models.py:restore_model_state_from_jaxnerf (when /root/reference is present),
through a stand-in `flax.training.checkpoints` module that only wraps our msgpack reader; the torch
NerfModel it fills must reproduce the oracle's eval_points_raw on the same parameters."""
import os
import sys
import types

import numpy as np
import pytest
import torch

from oracle import nerf_oracle as O
from plenoctree_amd import _lib
from plenoctree_amd.nerf_sh.nerf import checkpoints

REF = "/root/reference"


class _State:
    """TrainState stand-in without GPU kernels (repack is a no-op)."""

    def __init__(self, cfg, flat, step=0):
        self.cfg, self.params, self.step = cfg, flat.clone(), step
        self.m = torch.zeros_like(flat); self.v = torch.zeros_like(flat)

    def repack(self):
        pass


def _flat(deg, seed=3):
    ocfg = O.Cfg(sh_deg=deg)
    flat = O.flatten_params(O.init_params(ocfg, seed=seed))
    return ocfg, flat + 0.01 * torch.randn(flat.shape, generator=torch.Generator().manual_seed(seed))


@pytest.mark.parametrize("deg", [3, 4])
def test_roundtrip_and_tree_layout(tmp_path, deg):
    ocfg, flat = _flat(deg)
    cfg = _lib.make_cfg(sh_deg=deg)
    st = _State(cfg, flat, step=1234)
    st.m = torch.rand_like(flat); st.v = torch.rand_like(flat)
    path = checkpoints.save_checkpoint(str(tmp_path), st, 1234)
    assert os.path.basename(path) == "checkpoint_1234"
    raw = checkpoints.restore_checkpoint(str(tmp_path))            # target=None -> raw state dict
    params = raw["optimizer"]["target"]["params"]
    assert sorted(params) == ["MLP_0", "MLP_1"] and sorted(params["MLP_0"]) == [f"Dense_{i}" for i in range(10)]
    shapes = O.layer_shapes(ocfg)
    oparams = O.unflatten_params(flat, ocfg)
    for mi in range(2):
        for li, (fi, fo) in enumerate(shapes):
            k = params[f"MLP_{mi}"][f"Dense_{li}"]["kernel"]
            assert k.shape == (fi, fo) and k.dtype == np.float32
            np.testing.assert_array_equal(k, oparams[mi][li][0].numpy())
            np.testing.assert_array_equal(params[f"MLP_{mi}"][f"Dense_{li}"]["bias"], oparams[mi][li][1].numpy())
    assert int(raw["optimizer"]["state"]["step"]) == 1234
    ps = raw["optimizer"]["state"]["param_states"]["params"]["MLP_1"]["Dense_5"]["kernel"]
    assert sorted(ps) == ["grad_ema", "grad_sq_ema"] and ps["grad_ema"].shape == (319, 256)
    fresh = _State(cfg, torch.zeros_like(flat))
    assert checkpoints.restore_checkpoint(str(tmp_path), fresh) == path
    assert torch.equal(fresh.params, st.params) and torch.equal(fresh.m, st.m) and torch.equal(fresh.v, st.v)
    assert fresh.step == 1234
    # keep=N retention
    for s in (2000, 3000):
        checkpoints.save_checkpoint(str(tmp_path), st, s, keep=2)
    assert sorted(os.listdir(tmp_path)) == ["checkpoint_2000", "checkpoint_3000"]
    bad = _State(_lib.make_cfg(sh_deg=7 - deg), torch.zeros(2 * {3: 512588, 4: 505649}[deg]))
    with pytest.raises(ValueError):
        checkpoints.restore_checkpoint(str(tmp_path), bad)


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present (GPU box)")
def test_reference_consumer_reads_our_checkpoint(tmp_path, monkeypatch):
    ocfg, flat = _flat(3, seed=11)
    st = _State(_lib.make_cfg(sh_deg=3), flat, step=77)
    checkpoints.save_checkpoint(str(tmp_path), st, 77)
    # stand-in for the (uninstalled) flax package: only the restore entry point the reference calls
    flax = types.ModuleType("flax"); training = types.ModuleType("flax.training")
    ck = types.ModuleType("flax.training.checkpoints")
    ck.restore_checkpoint = lambda train_dir, target=None: checkpoints.restore_checkpoint(train_dir)
    flax.training = training; training.checkpoints = ck
    monkeypatch.setitem(sys.modules, "flax", flax)
    monkeypatch.setitem(sys.modules, "flax.training", training)
    monkeypatch.setitem(sys.modules, "flax.training.checkpoints", ck)
    monkeypatch.syspath_prepend(REF)
    from octree.nerf import models as ref_models        # the reference's torch twin
    model = ref_models.NerfModel(num_coarse_samples=64, num_fine_samples=128, use_viewdirs=False, sh_deg=3,
                                 sg_dim=-1, num_rgb_channels=48, num_sigma_channels=1)
    args = types.SimpleNamespace(train_dir=str(tmp_path))
    model = ref_models.restore_model_state_from_jaxnerf(args, model)     # octree/nerf/models.py:66-113
    pts = (torch.rand(50, 3, generator=torch.Generator().manual_seed(1)) * 2 - 1) * 1.5
    with torch.no_grad():
        rgb, sigma = model.eval_points_raw(pts)
        rgb_c, sigma_c = model.eval_points_raw(pts, coarse=True)
    params = O.unflatten_params(flat, ocfg)
    o_rgb, o_sigma = O.eval_points_raw(params, pts, ocfg)
    o_rgb_c, o_sigma_c = O.eval_points_raw(params, pts, ocfg, coarse=True)
    np.testing.assert_allclose(rgb.numpy(), o_rgb.numpy(), atol=2e-6)
    np.testing.assert_allclose(sigma.numpy(), o_sigma.numpy(), atol=2e-6)
    np.testing.assert_allclose(rgb_c.numpy(), o_rgb_c.numpy(), atol=2e-6)
    np.testing.assert_allclose(sigma_c.numpy(), o_sigma_c.numpy(), atol=2e-6)


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present (GPU box)")
def test_torch_state_dict_checkpoint_of_the_reference_twin(tmp_path, monkeypatch):
    """The reference's OTHER input format (octree/nerf/models.py:52-63, taken without --is_jaxnerf_ckpt): `*.ckpt` =
    {"model": state_dict of its torch NerfModel}.  The file is written by the reference's own module here; our reader must
    fill the arena so that the oracle reproduces the twin's eval_points_raw, and extraction's loader must pick the format the
    way the reference's flag does -- and refuse to go on with random weights when there is nothing to read."""
    monkeypatch.syspath_prepend(REF)
    from octree.nerf import models as ref_models
    torch.manual_seed(5)
    model = ref_models.NerfModel(num_coarse_samples=64, num_fine_samples=128, use_viewdirs=False, sh_deg=3,
                                 sg_dim=-1, num_rgb_channels=48, num_sigma_channels=1)
    with torch.no_grad():
        for p in model.parameters():            # biases are zero-initialised: make every leaf distinguishable
            p.add_(0.01 * torch.randn_like(p))
    torch.save({"model": model.state_dict()}, str(tmp_path / "000100.ckpt"))
    torch.save({"model": {k: torch.zeros_like(v) for k, v in model.state_dict().items()}}, str(tmp_path / "000050.ckpt"))
    ocfg = O.Cfg(sh_deg=3)
    st = _State(_lib.make_cfg(sh_deg=3), torch.zeros(2 * 505649))
    st.m += 1.0
    path = checkpoints.restore_torch_checkpoint(str(tmp_path), st)
    assert os.path.basename(path) == "000100.ckpt" and float(st.m.abs().max()) == 0.0        # sorted()[-1]; no stale moments
    # a file that also holds non-tensor objects (the reference's plain torch.load would unpickle them): refused with a message
    # that names the way out, read with trust_pickle=True
    import argparse
    extra = tmp_path / "extra"; extra.mkdir()
    torch.save({"model": model.state_dict(), "args": argparse.Namespace(lr=5e-4)}, str(extra / "000200.ckpt"))
    with pytest.raises(ValueError, match="trust_pickle=True"):
        checkpoints.restore_torch_checkpoint(str(extra), st)
    p0 = st.params.clone()
    assert checkpoints.restore_torch_checkpoint(str(extra), st, trust_pickle=True).endswith("000200.ckpt")
    assert torch.equal(st.params, p0)
    pts = (torch.rand(64, 3, generator=torch.Generator().manual_seed(2)) * 2 - 1) * 1.5
    with torch.no_grad():
        rgb, sigma = model.eval_points_raw(pts)
        rgb_c, sigma_c = model.eval_points_raw(pts, coarse=True)
    params = O.unflatten_params(st.params, ocfg)
    o_rgb, o_sigma = O.eval_points_raw(params, pts, ocfg)
    o_rgb_c, o_sigma_c = O.eval_points_raw(params, pts, ocfg, coarse=True)
    np.testing.assert_allclose(rgb.numpy(), o_rgb.numpy(), atol=2e-6)
    np.testing.assert_allclose(sigma.numpy(), o_sigma.numpy(), atol=2e-6)
    np.testing.assert_allclose(rgb_c.numpy(), o_rgb_c.numpy(), atol=2e-6)
    np.testing.assert_allclose(sigma_c.numpy(), o_sigma_c.numpy(), atol=2e-6)
    # extraction's loader: format chosen as the reference's flag does
    from plenoctree_amd.octree import extraction
    a = types.SimpleNamespace(train_dir=str(tmp_path), is_jaxnerf_ckpt=False)
    st2 = _State(_lib.make_cfg(sh_deg=3), torch.zeros(2 * 505649))
    assert "torch state dict" in extraction.load_nerf_checkpoint(a, st2) and torch.equal(st2.params, st.params)
    a.is_jaxnerf_ckpt = True
    with pytest.raises(FileNotFoundError):
        extraction.load_nerf_checkpoint(a, st2)                       # the flag asks for flax; only *.ckpt here
    checkpoints.save_checkpoint(str(tmp_path), _State(_lib.make_cfg(sh_deg=3), st.params + 1.0, step=9), 9)
    assert "flax msgpack" in extraction.load_nerf_checkpoint(a, st2) and torch.equal(st2.params, st.params + 1.0)
    empty = tmp_path / "empty"; empty.mkdir()
    a = types.SimpleNamespace(train_dir=str(empty), is_jaxnerf_ckpt=False)
    with pytest.raises(FileNotFoundError):
        extraction.load_nerf_checkpoint(a, st2)
    # a view-conditioned twin is rejected, not half-loaded
    with pytest.raises(ValueError, match="view-conditioned"):
        checkpoints.torch_state_dict_to_tree({"MLP_0.bottleneck_layer.weight": torch.zeros(1)})
