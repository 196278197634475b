"""Octree file compression for the in-browser viewer (reference: octree/compression.py): drops the keys a viewer does
not need, optionally replaces the SH coefficients by a median-cut palette per basis function, and deflates.

    python -m plenoctree_amd.octree.compression tree_opt.npz --out_dir min/ --overwrite [--noquant] [--bits 16]
        [--sigma_thresh 2.0] [--retain 0] [--weighted]

Output keys (what volrend and `N3Tree.load` read): `data_dim, child, invradius3, offset, data_format` plus either `data`
(--noquant) or `quant_colors [K', 2^bits, 3] f16, quant_map [K', n,2,2,2] u16, sigma [n,2,2,2]` and, with --retain r,
`data_retained [r, n,2,2,2,3] f16` for the first r basis functions (K' = K - r).  The palette comes from
`svox._quantize_median_cut` (torch; runs on the GPU when one is present - plumbing, not a HIP kernel: this tool is off
the hot path).
"""
import argparse
import os
import sys

import numpy as np
import torch

from . import svox

_DROPPED = ("parent_depth", "geom_resize_fact", "n_free", "n_internal", "depth_limit")     # compression.py:76-81


def define_flags():
    """compression.py:41-56."""
    p = argparse.ArgumentParser()
    p.add_argument("input", type=str, nargs="+", help="input npz(s)")
    p.add_argument("--noquant", action="store_true", help="disable quantization")
    p.add_argument("--bits", type=int, default=16, help="quantization bits (order)")
    p.add_argument("--out_dir", type=str, default="min_alt", help="where to write the compressed npz")
    p.add_argument("--overwrite", action="store_true", help="overwrite an existing compressed npz")
    p.add_argument("--weighted", action="store_true", help="weighted median cut")
    p.add_argument("--sigma_thresh", type=float, default=2.0, help="kill voxels under this sigma")
    p.add_argument("--retain", type=int, default=0, help="do not compress the first x SH coefficients")
    return p


def quantize(z, bits, sigma_thresh, retain=0, weighted=False, device=None):
    """compression.py:88-136 on the dict of a loaded tree file (modified in place): `data` -> palette form."""
    dev = device or ("cuda" if torch.cuda.is_available() else "cpu")
    data = torch.from_numpy(np.asarray(z["data"]))                 # float16 on disk; sigma stays in the file's dtype
    cells = data.shape[:-1]                                        # (n, N, N, N)
    sigma = data[..., -1].reshape(-1).clone()
    alive = sigma > sigma_thresh
    sigma[~alive] = 0.0
    K = (data.shape[-1] - 1) // 3
    rgb = data[..., :-1].reshape(-1, 3, K).float()[alive]          # [m, channel, basis]
    weights = (1.0 - torch.exp(-0.01 * sigma[alive].float())) if weighted else torch.empty(0)
    alive_np = alive.numpy()
    if retain:
        kept = np.zeros((retain, alive_np.shape[0], 3), np.float16)
        kept[:, alive_np] = rgb[..., :retain].permute(2, 0, 1).numpy().astype(np.float16)
        z["data_retained"] = kept.reshape((retain,) + tuple(cells) + (3,))
    palettes, maps = [], []
    for b in range(retain, K):
        colors, ids = svox._quantize_median_cut(rgb[..., b].contiguous().to(dev), weights.to(dev), bits)
        full = np.zeros(alive_np.shape[0], np.uint16)
        full[alive_np] = ids.cpu().numpy().astype(np.uint16)
        palettes.append(colors.cpu().numpy().astype(np.float16))
        maps.append(full.reshape(cells))
    z["quant_colors"] = np.stack(palettes, 0)
    z["quant_map"] = np.stack(maps, 0)
    z["sigma"] = sigma.reshape(cells).numpy()
    del z["data"]
    return z


def compress_file(src, dst, args):
    z = np.load(src)
    if not args.noquant and "quant_colors" in z.files:
        print(" > skip since source already compressed")
        return False
    z = {k: z[k] for k in z.files if k not in _DROPPED}
    if not args.noquant:
        if args.bits < 1 or args.bits > 16:
            raise ValueError("--bits must be in 1..16 (the map is stored as uint16)")
        quantize(z, args.bits, args.sigma_thresh, args.retain, args.weighted)
    np.savez_compressed(dst, **z)
    print(" > Size", os.path.getsize(src) // (1024 * 1024), "MB ->", os.path.getsize(dst) // (1024 * 1024), "MB")
    return True


@torch.no_grad()
def main(argv=None):
    args = define_flags().parse_args(argv)
    os.makedirs(args.out_dir, exist_ok=True)
    print("Quantization disabled, only applying deflate" if args.noquant else "Quantization enabled")
    done = []
    for src in args.input:
        dst = os.path.join(args.out_dir, os.path.basename(src))
        print("Compressing", src, "to", dst)
        if not args.overwrite and os.path.exists(dst):
            print(" > skip")
            continue
        if compress_file(src, dst, args):
            done.append(dst)
    return done


if __name__ == "__main__":
    main(sys.argv[1:])
