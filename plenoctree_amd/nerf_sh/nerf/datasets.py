"""Ray/pixel feeders for the NeRF-SH trainer.

`Blender` and `NSVF` follow the reference's loaders (nerf_sh/nerf/datasets.py:189-232: transforms_*.json,
RGBA composited on white, focal from camera_angle_x; :491-552: intrinsics.txt + pose/ + rgb/) and batch sampler (:149-167: one random
image, B random pixels with replacement).  `Synthetic` is the stand-in used where no dataset
exists (the GPU box has none): the same Blender camera convention and sampler over an
analytic scene whose pixel colours are a closed-form function of the ray, so targets are
identical for the oracle and the HIP path and PSNR is meaningful."""
import json
import os

import numpy as np
import torch

from . import utils


def pose_spherical(theta_deg, phi_deg, radius):
    """Blender-convention camera-to-world looking at the origin (-z forward)."""
    th, ph = np.deg2rad(theta_deg), np.deg2rad(phi_deg)
    eye = radius * np.array([np.cos(ph) * np.sin(th), np.cos(ph) * np.cos(th), np.sin(ph)], np.float64)
    fwd = -eye / np.linalg.norm(eye)
    up = np.array([0.0, 0.0, 1.0])
    right = np.cross(fwd, up); right /= np.linalg.norm(right)
    upv = np.cross(right, fwd)
    c2w = np.eye(4, dtype=np.float32)
    c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = right, upv, -fwd, eye
    return c2w


def area_resize(image, out_h, out_w):
    """cv2.resize(image, (out_w, out_h), interpolation=cv2.INTER_AREA) for down-scaling a float [H,W,C] image (cv2 is
    not installed): every output pixel is the area-weighted mean of the source pixels its footprint [j s, (j+1) s)
    covers, s = in / out, separable.  For an integer factor on a divisible size (the reference's factor = 2 on the
    800 x 800 Blender images, nerf_sh/nerf/datasets.py:208-212) this is exactly the mean of factor x factor blocks."""
    def weights(n_in, n_out):
        s = n_in / n_out
        w = np.zeros((n_out, n_in), np.float64)
        for j in range(n_out):
            lo, hi = j * s, (j + 1) * s
            for i in range(int(np.floor(lo)), min(int(np.ceil(hi)), n_in)):
                w[j, i] = max(0.0, min(hi, i + 1) - max(lo, i)) / s
        return w
    img = np.asarray(image, np.float32)
    wy, wx = weights(img.shape[0], out_h), weights(img.shape[1], out_w)
    # separable: rows first, then columns (two matrix products; a three-operand einsum without a contraction path is
    # one loop over y*x*i*j*c -- minutes per 800 x 800 image)
    rows = np.tensordot(wy, img.astype(np.float64), axes=(1, 0))          # [out_h, W, C]
    return np.einsum("yjc,xj->yxc", rows, wx).astype(np.float32)


def analytic_scene_rgb(origins, directions, white_bkgd=True, return_alpha=False):
    """Colour seen along each ray: three hard spheres with view-dependent shading.  return_alpha: also the hit mask (1.0 where a
    sphere is hit, 0.0 on the background) -- the alpha channel an RGBA export of the scene carries (scripts/export_scene.py)."""
    o, d = origins.double(), directions.double()
    d = d / d.norm(dim=-1, keepdim=True)
    centers = torch.tensor([[0.0, 0.0, 0.0], [0.7, 0.3, 0.2], [-0.5, -0.4, 0.5]], dtype=torch.float64, device=o.device)
    radii = torch.tensor([0.6, 0.3, 0.35], dtype=torch.float64, device=o.device)
    base = torch.tensor([[0.8, 0.3, 0.2], [0.2, 0.7, 0.3], [0.2, 0.3, 0.9]], dtype=torch.float64, device=o.device)
    best_t = torch.full(o.shape[:-1], float("inf"), dtype=torch.float64, device=o.device)
    color = torch.ones(*o.shape[:-1], 3, dtype=torch.float64, device=o.device) * (1.0 if white_bkgd else 0.0)
    for c, r, b in zip(centers, radii, base):
        oc = o - c
        bq = (oc * d).sum(-1)
        disc = bq * bq - ((oc * oc).sum(-1) - r * r)
        hit = disc > 0
        t = -bq - torch.sqrt(disc.clamp(min=0))
        hit = hit & (t > 0) & (t < best_t)
        n = (o + t[..., None] * d - c) / r
        diffuse = (0.35 + 0.65 * (n * torch.tensor([0.3, 0.5, 0.8], dtype=torch.float64, device=o.device)).sum(-1).clamp(min=0))
        spec = (-(n * d).sum(-1)).clamp(min=0) ** 8
        col = (b * diffuse[..., None] + 0.3 * spec[..., None]).clamp(0, 1)
        color = torch.where(hit[..., None], col, color)
        best_t = torch.where(hit, t, best_t)
    if return_alpha:
        return color.float(), torch.isfinite(best_t).float()
    return color.float()


class HipFeeder:
    """Batch sampling and ray generation on the MI355X (pxo_randint, pxo_generate_rays[_multi]): no host->device
    copy inside the step loop, so the host never waits for the GPU and launches run ahead of the kernels.  This is the
    only feeder of the product; the CPU host-logic tests install their own (tests/_cpu_feeder.py) through
    `Dataset.feeder_factory`."""
    resident_images = True        # Synthetic renders its training views once and keeps them on the device

    def __init__(self, device):
        if device.type != "cuda":
            raise RuntimeError("datasets: the ray/pixel feeder runs on a ROCm device only (no CPU fallback)")
        from ... import ops
        self.ops, self.device = ops, device

    def randint(self, seed, draw, count, n):
        """`count` ids uniform in [0, n) from Philox stream `draw` of `seed` (np.random.randint, datasets.py:160-166)."""
        return self.ops.randint(seed, draw, count, n, device=self.device)

    def generate_rays(self, c2w, w, h, focal, ray_indices):
        return self.ops.generate_rays(c2w, w, h, focal, ray_indices)

    def generate_rays_multi(self, c2w_all, w, h, focal, ray_ids):
        return self.ops.generate_rays_multi(c2w_all, w, h, focal, ray_ids)

    def sample_batch(self, seed, draw, count, c2w, w, h, focal, image_rgb, first=0):
        """randint + generate_rays + the gather of the image's colours in one launch (pxo_sample_batch): the same values as
        the three calls above, bit for bit.  `first`: rows = elements first .. first + count - 1 of the draw."""
        return self.ops.sample_batch(seed, draw, c2w, w, h, focal, image_rgb, count, first=first)


class Dataset:
    """Iterator yielding {"pixels": [B,3], "rays": Rays([B,3] x3)} on `device`."""
    feeder_factory = HipFeeder

    def __init__(self, split, args, device, batch_size=None, seed=20201473, shard=None):
        # shard = (rank, world): the reference's SINGLE-HOST step -- one image per step for the whole host, its global batch
        # of pixels drawn once and cut into `world` contiguous pieces (datasets.py:159-166 + utils.shard, utils.py:518-522).
        # Every rank then carries the same `seed` (same image sequence, same pixel draw) and `batch_size` is the rank's piece.
        # Without it each rank is its own host (seed + host_id, train.py:128): its own image, its own pixels.
        self.shard = None if shard is None else (int(shard[0]), int(shard[1]))
        if self.shard is not None and not (0 <= self.shard[0] < self.shard[1]):
            raise ValueError(f"shard {shard}: rank must be in [0, world)")
        self.split = split
        self.device = device
        self.feeder = type(self).feeder_factory(device)
        self.batch_size = batch_size if batch_size is not None else args.batch_size
        self.white_bkgd = bool(args.white_bkgd)
        self.image_batching = bool(getattr(args, "image_batching", False))
        self.rng = np.random.RandomState(seed)      # np.random.seed(20201473 + host_id), train.py:128
        self.seed, self.draws = int(seed), 0
        self.it = 0
        self._load(args)
        self._c2w_dev = torch.from_numpy(np.ascontiguousarray(self.camtoworlds[:, :3, :4])).to(self.device)

    # subclasses set: self.camtoworlds [n,4,4] float32, self.h, self.w, self.focal, self.n_examples
    def _load(self, args):
        raise NotImplementedError

    def _pixels_for(self, image_index, ray_indices, rays):
        raise NotImplementedError

    @property
    def size(self):
        return self.n_examples

    def _rays_for(self, image_index, ray_indices):
        """Rays of the chosen pixels of one image (generate_rays, utils.py:545-589), on device."""
        c2w = torch.from_numpy(np.ascontiguousarray(self.camtoworlds[image_index, :3, :4])).to(self.device) \
            if getattr(self, "_c2w_dev", None) is None else self._c2w_dev[image_index]
        return utils.Rays(*self.feeder.generate_rays(c2w, self.w, self.h, self.focal, ray_indices))

    def __iter__(self):
        return self

    def __next__(self):
        if self.split == "train" and self.image_batching:
            if self.shard is not None:
                raise ValueError("shard=(rank, world) is the single-image sampler's option (image_batching draws from all images)")
            return self._next_train_all_images()
        if self.split == "train":
            # datasets.py:159-166: one random image, batch_size random pixels (with replacement)
            image_index = int(self.rng.randint(0, self.n_examples))
            self.draws += 1
            rank, world = self.shard if self.shard is not None else (0, 1)
            first = rank * self.batch_size
            if getattr(self, "images", None) is not None and hasattr(self.feeder, "sample_batch"):
                # resident images on the device: ids, rays and colours in one launch
                o, d, v, px = self.feeder.sample_batch(self.seed, self.draws, self.batch_size, self._c2w_dev[image_index],
                                                       self.w, self.h, self.focal, self.images[image_index], first=first)
                return {"pixels": px, "rays": utils.Rays(o, d, v)}
            ray_indices = self.feeder.randint(self.seed, self.draws, self.batch_size * world, self.h * self.w)
            if world > 1:
                ray_indices = ray_indices[first:first + self.batch_size].contiguous()
            rays = self._rays_for(image_index, ray_indices)
            return {"pixels": self._pixels_for(image_index, ray_indices, rays), "rays": rays}
        idx = self.it
        self.it = (self.it + 1) % self.n_examples
        return self.get_image(idx)

    def _next_train_all_images(self):
        """image_batching (datasets.py:137-141,152-157): batch_size rays drawn from the flattened table of the rays
        of ALL images.  The table itself is never materialised: a ray id is (camera, pixel) and the ray is computed."""
        hw = self.h * self.w
        self.draws += 1
        ids = self.feeder.randint(self.seed, self.draws, self.batch_size, self.n_examples * hw)
        rays = utils.Rays(*self.feeder.generate_rays_multi(self._c2w_dev, self.w, self.h, self.focal, ids))
        return {"pixels": self._pixels_flat(ids, rays), "rays": rays}

    def _pixels_flat(self, ids, rays):
        """Target colours of ids into the flattened [n_examples * H*W] pixel table."""
        if getattr(self, "images", None) is not None:
            return self.images.reshape(-1, 3)[ids].contiguous()
        return self._render(None, None, rays)

    def get_image(self, idx):
        ray_indices = torch.arange(self.h * self.w, device=self.device)
        rays = self._rays_for(idx, ray_indices)
        px = self._pixels_for(idx, ray_indices, rays)
        rs = utils.namedtuple_map(lambda r: r.reshape(self.h, self.w, 3), rays)
        return {"pixels": px.reshape(self.h, self.w, 3), "rays": rs}

    def peek(self):
        return self.get_image(self.it) if self.split != "train" else next(self)


class Synthetic(Dataset):
    """100 train / 200 test Blender-convention poses on a sphere of radius 4.0311 around an
    analytic scene; 800x800, camera_angle_x = 0.6911112.  Like the reference's loaders
    (datasets.py:189-232: all images decoded once at start-up) the training images are rendered once at
    construction and kept resident (100 x 800 x 800 x 3 f32 = 0.77 GB on the device); a batch is then one gather."""

    def _load(self, args):
        # Optional attributes of `args` (not flags; scripts/export_scene.py and the on-disk-format tests set them): synthetic_hw =
        # (h, w) with the same horizontal field of view, synthetic_views = (n_train, n_test), synthetic_8bit = colours rounded to
        # k / 255 -- what a PNG of the view holds, so that a run on the exported files and a run on this class see the same bits.
        n_views = getattr(args, "synthetic_views", None) or (100, 200)
        n = n_views[0] if self.split == "train" else n_views[1]
        side = 800 if args.factor == 0 else max(800 // args.factor, 8)
        self.h, self.w = getattr(args, "synthetic_hw", None) or (side, side)
        self.focal = 0.5 * self.w / np.tan(0.5 * 0.6911112)
        self.quantize8 = bool(getattr(args, "synthetic_8bit", False))
        rs = np.random.RandomState(7 if self.split == "train" else 11)
        self.camtoworlds = np.stack([pose_spherical(rs.uniform(0, 360), rs.uniform(-10, 60), 4.0311) for _ in range(n)])
        self.n_examples = n
        self.images = None
        if self.split == "train" and self.feeder.resident_images:
            ids = torch.arange(self.h * self.w, device=self.device)
            chunks = []
            for i0 in range(0, n, 20):                     # 20 views per evaluation of the analytic scene
                rays = [self._rays_for(i, ids) for i in range(i0, min(i0 + 20, n))]
                o = torch.cat([r.origins for r in rays]); d = torch.cat([r.directions for r in rays])
                chunks.append(self._q8(analytic_scene_rgb(o, d, self.white_bkgd)).reshape(len(rays), self.h * self.w, 3))
            self.images = torch.cat(chunks).contiguous()

    def _q8(self, rgb):
        if not self.quantize8:
            return rgb
        # k / 255 exactly as the loaders compute it (numpy float32 division of the decoded byte): a table, because torch divides
        # by a scalar as a multiplication by its reciprocal on the device, which is 1 ulp off for some k
        lut = torch.from_numpy(np.arange(256, dtype=np.float32) / np.float32(255.0)).to(rgb.device)
        return lut[torch.round(rgb * 255.0).long().clamp_(0, 255)]

    def _render(self, image_index, ray_indices, rays):
        return self._q8(analytic_scene_rgb(rays.origins, rays.directions, self.white_bkgd)).contiguous()

    def _pixels_for(self, image_index, ray_indices, rays):
        if self.images is not None:
            return self.images[image_index][ray_indices].contiguous()
        return self._render(image_index, ray_indices, rays)


class Blender(Dataset):
    """NeRF-Synthetic loader (reference nerf_sh/nerf/datasets.py:189-232)."""

    def _load(self, args):
        from PIL import Image
        with open(os.path.join(args.data_dir, f"transforms_{self.split}.json"), "r") as fp:
            meta = json.load(fp)
        images, cams = [], []
        for frame in meta["frames"]:
            fname = os.path.join(args.data_dir, frame["file_path"] + ".png")
            image = np.asarray(Image.open(fname), dtype=np.float32) / 255.0
            if args.factor == 2:            # datasets.py:208-212: the float RGBA image, before compositing
                image = area_resize(image, image.shape[0] // 2, image.shape[1] // 2)
            elif args.factor > 0:           # :213-216
                raise ValueError(f"Blender dataset only supports factor=0 or 2, {args.factor} set.")
            cams.append(np.array(frame["transform_matrix"], dtype=np.float32))
            images.append(image)
        images = np.stack(images, 0)
        if self.white_bkgd:
            images = images[..., :3] * images[..., -1:] + (1.0 - images[..., -1:])   # datasets.py:219-222
        else:
            images = images[..., :3]
        self.h, self.w = images.shape[1:3]
        self.camtoworlds = np.stack(cams, 0)
        self.focal = 0.5 * self.w / np.tan(0.5 * float(meta["camera_angle_x"]))      # :226-228
        self.n_examples = images.shape[0]
        # resident on the device (100 x 800 x 800 x 3 f32 = 0.77 GB of 288): the per-batch gather is one kernel
        self.images = torch.from_numpy(images.reshape(self.n_examples, -1, 3)).to(self.device)

    def _pixels_for(self, image_index, ray_indices, rays):
        return self.images[image_index][ray_indices].contiguous()


class NSVF(Dataset):
    """NSVF-format loader used by the Tanks&Temples preset (reference nerf_sh/nerf/datasets.py:491-552;
    bbox.txt as in octree/nerf/datasets.py:73-77): intrinsics.txt, pose/<split>_*.txt, rgb/<split>_*.png
    with split prefix 0_ train, 1_ val, 2_ test (falling back to 1_ when there is no 2_)."""

    def _load(self, args):
        from PIL import Image
        root = os.path.expanduser(args.data_dir)
        K = np.loadtxt(os.path.join(root, "intrinsics.txt"))
        pose_files = sorted(os.listdir(os.path.join(root, "pose")))
        img_files = sorted(os.listdir(os.path.join(root, "rgb")))
        prefix = {"train": "0_", "val": "1_", "test": "2_"}[self.split]
        if self.split == "test" and not any(f.startswith("2_") for f in pose_files):
            prefix = "1_"
        pose_files = [f for f in pose_files if f.startswith(prefix)]
        img_files = [f for f in img_files if f.startswith(prefix)]
        if len(pose_files) != len(img_files):
            raise ValueError(f"NSVF {root}: {len(img_files)} images but {len(pose_files)} poses for split {self.split}")
        cam_trans = np.diag(np.array([1, -1, -1, 1], dtype=np.float32))       # OpenCV -> OpenGL axes (:517)
        images, cams = [], []
        for img_name, pose_name in zip(img_files, pose_files):
            image = np.asarray(Image.open(os.path.join(root, "rgb", img_name)), dtype=np.float32) / 255.0
            cams.append(np.loadtxt(os.path.join(root, "pose", pose_name)) @ cam_trans)
            if image.shape[-1] == 4:
                image = image[..., :3] * image[..., -1:] + (1.0 - image[..., -1:]) if self.white_bkgd else image[..., :3]
            if args.factor > 1:             # :430-434: after compositing, any integer factor
                image = area_resize(image, image.shape[0] // args.factor, image.shape[1] // args.factor)
            images.append(image)
        images = np.stack(images, 0)
        self.n_examples, self.h, self.w = images.shape[:3]
        self.camtoworlds = np.stack(cams, 0).astype(np.float32)
        self.focal = float(K[0, 0] + K[1, 1]) * 0.5                            # :548-551
        if args.factor > 1:
            self.focal /= args.factor
        bbox_path = os.path.join(root, "bbox.txt")
        self.bbox = np.loadtxt(bbox_path)[:-1] if os.path.isfile(bbox_path) else None
        self.images = torch.from_numpy(images.reshape(self.n_examples, -1, 3)).to(self.device)

    def _pixels_for(self, image_index, ray_indices, rays):
        return self.images[image_index][ray_indices].contiguous()


dataset_dict = {"blender": Blender, "nsvf": NSVF, "synthetic": Synthetic}


def get_dataset(split, args, device, batch_size=None, seed=20201473, shard=None):
    if args.dataset not in dataset_dict:
        raise NotImplementedError(f"dataset {args.dataset} is not built on the MI355X path")
    return dataset_dict[args.dataset](split, args, device, batch_size=batch_size, seed=seed, shard=shard)
