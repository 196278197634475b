"""Torch-CPU stand-in for plenoctree_amd.nerf_sh.nerf.datasets.HipFeeder (TEST INFRASTRUCTURE: the product has no
CPU path).  The host-logic tests (`-m "not gpu"`) run the loaders / samplers / drivers with it; conftest.py installs
it through `Dataset.feeder_factory` for datasets built on a CPU device.

generate_rays follows nerf_sh/nerf/utils.py:545-589 (pinhole, integer pixel centres); the sampler is the legacy
numpy generator the reference uses (np.random.randint, nerf_sh/nerf/datasets.py:160-166)."""
import numpy as np
import torch


class CpuFeeder:
    resident_images = False

    def __init__(self, device):
        self.device = device
        self._rng = {}

    def randint(self, seed, draw, count, n):
        rng = self._rng.setdefault(seed, np.random.RandomState(seed))
        return torch.from_numpy(rng.randint(0, n, (count,))).to(self.device)

    def generate_rays(self, c2w, w, h, focal, ray_indices):
        idx = ray_indices
        x = (idx % w).float()
        y = torch.div(idx, w, rounding_mode="floor").float()
        cam = torch.stack([(x - w * 0.5) / focal, -(y - h * 0.5) / focal, -torch.ones_like(x)], -1)
        directions = cam @ c2w[:3, :3].T
        origins = c2w[:3, 3].expand_as(directions).contiguous()
        viewdirs = directions / directions.norm(dim=-1, keepdim=True)
        return origins, directions.contiguous(), viewdirs.contiguous()

    def generate_rays_multi(self, c2w_all, w, h, focal, ray_ids):
        hw = h * w
        cam = torch.div(ray_ids, hw, rounding_mode="floor")
        o = torch.empty(ray_ids.shape[0], 3); d = torch.empty_like(o); v = torch.empty_like(o)
        for c in torch.unique(cam).tolist():
            sel = cam == c
            o[sel], d[sel], v[sel] = self.generate_rays(c2w_all[int(c)], w, h, focal, ray_ids[sel] - int(c) * hw)
        return o, d, v


def feeder_for(device):
    """HipFeeder on a ROCm device, CpuFeeder on the CPU (tests only)."""
    if device.type == "cuda":
        from plenoctree_amd.nerf_sh.nerf.datasets import HipFeeder
        return HipFeeder(device)
    return CpuFeeder(device)
