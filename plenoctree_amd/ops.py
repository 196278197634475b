"""torch-tensor front end of the C ABI (include/plenoctree_hip.h).

torch is used for device storage and the current HIP stream only; all arithmetic happens in
libplenoctree_hip.so.  Every tensor must be a contiguous float32 ROCm tensor.
"""
import ctypes

import torch

from . import _lib
from ._lib import NET_DEPTH, NET_WIDTH, ENC_PAD, PxoError, check, make_cfg  # noqa: F401


def _require_gpu():
    if not torch.cuda.is_available():
        raise PxoError("plenoctree_amd needs a ROCm GPU (gfx950); there is no CPU fallback")


def _p(t):
    if t is None:
        return None
    if not (t.is_cuda and t.is_contiguous()):
        raise PxoError("tensor must be a contiguous ROCm tensor")
    return ctypes.c_void_p(t.data_ptr())


def _f(t):
    if t is not None and t.dtype != torch.float32:
        raise PxoError(f"expected float32, got {t.dtype}")
    return _p(t)


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _new(*shape, device=None, dtype=torch.float32):
    return torch.empty(*shape, dtype=dtype, device=device or torch.device("cuda", torch.cuda.current_device()))


def sh_dim(cfg):
    return (cfg.sh_deg + 1) ** 2


def rgb_channels(cfg):
    return 3 * sh_dim(cfg)


def param_layout(cfg):
    """[(layer, is_bias, offset, rows, cols)] of ONE MLP's sub-arena and its size in floats."""
    lib = _lib.load()
    leaves = (_lib.PxoLeaf * _lib.NUM_LEAVES)()
    n = ctypes.c_int64(0)
    check(lib.pxo_param_layout(ctypes.byref(cfg), leaves, ctypes.byref(n)), "pxo_param_layout")
    return [(l.layer, l.is_bias, l.offset, l.rows, l.cols) for l in leaves], n.value


def packed_sizes(cfg):
    lib = _lib.load()
    a, b = ctypes.c_int64(0), ctypes.c_int64(0)
    check(lib.pxo_packed_sizes(ctypes.byref(cfg), ctypes.byref(a), ctypes.byref(b)), "pxo_packed_sizes")
    return a.value, b.value


def pack_weights(cfg, mlp_params, packed_fwd=None, packed_bwd=None, need_bwd=True):
    _require_gpu()
    lib = _lib.load()
    nf, nb = packed_sizes(cfg)
    if packed_fwd is None:
        packed_fwd = _new(nf, device=mlp_params.device)
    if need_bwd and packed_bwd is None:
        packed_bwd = _new(nb, device=mlp_params.device)
    check(lib.pxo_pack_weights(ctypes.byref(cfg), _f(mlp_params), _f(packed_fwd),
                               _f(packed_bwd) if need_bwd else None, _stream()), "pxo_pack_weights")
    return packed_fwd, packed_bwd


def sample_along_rays(origins, directions, num_samples, near, far, t_rand=None, lindisp=False):
    _require_gpu()
    lib = _lib.load()
    B = origins.shape[0]
    z = _new(B, num_samples, device=origins.device)
    pts = _new(B, num_samples, 3, device=origins.device)
    check(lib.pxo_sample_along_rays(_f(origins), _f(directions), B, num_samples, near, far, int(lindisp),
                                    _f(t_rand), _f(z), _f(pts), _stream()), "pxo_sample_along_rays")
    return z, pts


def posenc(x):
    _require_gpu()
    lib = _lib.load()
    lead = x.shape[:-1]
    x2 = x.reshape(-1, 3).contiguous()
    enc = _new(x2.shape[0], _lib.ENC_DIM, device=x.device)
    check(lib.pxo_posenc(_f(x2), x2.shape[0], _f(enc), _stream()), "pxo_posenc")
    return enc.reshape(*lead, _lib.ENC_DIM)


def relu_mask_bytes(M):
    return _lib.load().pxo_relu_mask_bytes(M)


def dbias_partial_bytes(M):
    return _lib.load().pxo_dbias_partial_bytes(M)


def mlp_fwd(cfg, packed_fwd, pts, save=False, want_rgb=True):
    """-> raw_rgb [M,3K] (or None), raw_sigma [M], and if save: (acts [8,M,256], enc [M,64], mask)."""
    _require_gpu()
    lib = _lib.load()
    pts = pts.reshape(-1, 3)
    M = pts.shape[0]
    dev = pts.device
    raw_rgb = _new(M, rgb_channels(cfg), device=dev) if want_rgb else None
    raw_sigma = _new(M, device=dev)
    acts = enc = mask = None
    if save:
        acts = _new(NET_DEPTH, M, NET_WIDTH, device=dev)
        enc = _new(M, ENC_PAD, device=dev)
        mask = _new(max(relu_mask_bytes(M), 16), device=dev, dtype=torch.uint8)
    check(lib.pxo_mlp_fwd(ctypes.byref(cfg), _f(packed_fwd), _f(pts), M, _f(raw_rgb), _f(raw_sigma),
                          _f(acts), _f(enc), _p(mask), _stream()), "pxo_mlp_fwd")
    if save:
        return raw_rgb, raw_sigma, (acts, enc, mask)
    return raw_rgb, raw_sigma


def mlp_bwd_data(cfg, packed_bwd, d_raw_rgb, d_raw_sigma, mask):
    _require_gpu()
    lib = _lib.load()
    M = d_raw_sigma.shape[0]
    dev = d_raw_sigma.device
    dz = _new(NET_DEPTH, M, NET_WIDTH, device=dev)
    dbias = _new(max(dbias_partial_bytes(M) // 4, 4), device=dev)
    check(lib.pxo_mlp_bwd_data(ctypes.byref(cfg), _f(packed_bwd), _f(d_raw_rgb), _f(d_raw_sigma), _p(mask), M,
                               _f(dz), _f(dbias), _stream()), "pxo_mlp_bwd_data")
    return dz, dbias


def mlp_bwd_weights(cfg, acts, enc, dz, d_raw_rgb, d_raw_sigma, dbias):
    _require_gpu()
    lib = _lib.load()
    M = d_raw_sigma.shape[0]
    dev = d_raw_sigma.device
    _, n = param_layout(cfg)
    grads = _new(n, device=dev)
    nbytes = ctypes.c_size_t(0)
    check(lib.pxo_wgrad_workspace_bytes(ctypes.byref(cfg), M, ctypes.byref(nbytes)), "pxo_wgrad_workspace_bytes")
    ws = _new(max(nbytes.value, 16), device=dev, dtype=torch.uint8)
    check(lib.pxo_mlp_bwd_weights(ctypes.byref(cfg), _f(acts), _f(enc), _f(dz), _f(d_raw_rgb), _f(d_raw_sigma),
                                  _f(dbias), M, _f(grads), _p(ws), nbytes.value, _stream()),
          "pxo_mlp_bwd_weights")
    return grads


def shade_composite_fwd(cfg, raw_rgb, raw_sigma, z_vals, directions, viewdirs):
    _require_gpu()
    lib = _lib.load()
    B, S = z_vals.shape
    dev = z_vals.device
    comp, disp, acc, w = _new(B, 3, device=dev), _new(B, device=dev), _new(B, device=dev), _new(B, S, device=dev)
    check(lib.pxo_shade_composite_fwd(ctypes.byref(cfg), _f(raw_rgb), _f(raw_sigma), _f(z_vals), _f(directions),
                                      _f(viewdirs), B, S, _f(comp), _f(disp), _f(acc), _f(w), _stream()),
          "pxo_shade_composite_fwd")
    return comp, disp, acc, w


def shade_composite_bwd(cfg, raw_rgb, raw_sigma, z_vals, directions, viewdirs, d_comp_rgb):
    _require_gpu()
    lib = _lib.load()
    B, S = z_vals.shape
    dev = z_vals.device
    d_rgb = _new(B * S, rgb_channels(cfg), device=dev)
    d_sigma = _new(B * S, device=dev)
    check(lib.pxo_shade_composite_bwd(ctypes.byref(cfg), _f(raw_rgb), _f(raw_sigma), _f(z_vals), _f(directions),
                                      _f(viewdirs), _f(d_comp_rgb), B, S, _f(d_rgb), _f(d_sigma), _stream()),
          "pxo_shade_composite_bwd")
    return d_rgb, d_sigma


def sample_pdf(z_coarse, w_coarse, origins, directions, num_fine, u=None):
    _require_gpu()
    lib = _lib.load()
    B, Nc = z_coarse.shape
    dev = z_coarse.device
    z = _new(B, Nc + num_fine, device=dev)
    pts = _new(B, Nc + num_fine, 3, device=dev)
    check(lib.pxo_sample_pdf(_f(z_coarse), _f(w_coarse), _f(origins), _f(directions), B, Nc, num_fine, _f(u),
                             _f(z), _f(pts), _stream()), "pxo_sample_pdf")
    return z, pts


def add_gaussian_noise(raw, noise_std, noise=None, seed=0, stream_id=0):
    """model_utils.add_gaussian_noise in place: raw += noise_std * N(0,1) (`noise`: injected draws, else Philox)."""
    _require_gpu()
    check(_lib.load().pxo_add_gaussian_noise(_f(raw), raw.numel(), float(noise_std), _f(noise), seed, stream_id, _stream()),
          "pxo_add_gaussian_noise")
    return raw


def uniform(seed, stream_id, n, lo=0.0, hi=1.0, device=None):
    _require_gpu()
    lib = _lib.load()
    out = _new(n, device=device)
    check(lib.pxo_uniform(seed, stream_id, n, lo, hi, _f(out), _stream()), "pxo_uniform")
    return out


def adam_step(params, m, v, grads, lr, step, grad_scale=1.0):
    _require_gpu()
    lib = _lib.load()
    check(lib.pxo_adam_step(_f(params), _f(m), _f(v), _f(grads), params.numel(), float(lr), int(step),
                            float(grad_scale), _stream()), "pxo_adam_step")


def adam_pack_step(cfg, params, m, v, grads, lr, step, packed, grad_scale=1.0):
    """Adam on the whole arena + refresh of the weight images `packed` = [(fwd0, bwd0), (fwd1, bwd1)] in one launch."""
    _require_gpu()
    lib = _lib.load()
    (f0, b0), (f1, b1) = packed
    check(lib.pxo_adam_pack_step(ctypes.byref(cfg), _f(params), _f(m), _f(v), _f(grads), float(lr), int(step),
                                 float(grad_scale), _f(f0), _f(b0), _f(f1), _f(b1), _stream()), "pxo_adam_pack_step")


def shade_composite_train(cfg, raw_rgb, raw_sigma, z_vals, directions, viewdirs, pixels, n_sp=0, want_rgb=True,
                          want_weights=True):
    """Forward compositing + pixel loss + reverse in one launch: returns dict(comp_rgb, weights, ray_sse, d_raw_rgb,
    d_raw_sigma, sp_exp).  raw_rgb [B*S + n_sp, 3K], raw_sigma [B*S + n_sp]."""
    _require_gpu()
    lib = _lib.load()
    B, S = z_vals.shape
    C = raw_rgb.shape[-1]
    dev = raw_rgb.device
    out = {"comp_rgb": _new(B, 3, device=dev) if want_rgb else None,
           "weights": _new(B, S, device=dev) if want_weights else None,
           "ray_sse": _new(B, device=dev), "d_raw_rgb": _new(B * S + n_sp, C, device=dev),
           "d_raw_sigma": _new(B * S + n_sp, device=dev), "sp_exp": _new(max(n_sp, 1), device=dev)}
    check(lib.pxo_shade_composite_train(ctypes.byref(cfg), _f(raw_rgb), _f(raw_sigma), _f(z_vals), _f(directions),
                                        _f(viewdirs), _f(pixels), B, S, _f(out["comp_rgb"]), _f(out["weights"]),
                                        _f(out["ray_sse"]), _f(out["d_raw_rgb"]), _f(out["d_raw_sigma"]), n_sp,
                                        _f(out["sp_exp"]), _stream()), "pxo_shade_composite_train")
    return out


def render_workspace_bytes(cfg, B):
    n = ctypes.c_size_t(0)
    check(_lib.load().pxo_render_workspace_bytes(ctypes.byref(cfg), B, ctypes.byref(n)), "pxo_render_workspace_bytes")
    return n.value


def train_workspace_bytes(cfg, B):
    n = ctypes.c_size_t(0)
    check(_lib.load().pxo_train_workspace_bytes(ctypes.byref(cfg), B, ctypes.byref(n)), "pxo_train_workspace_bytes")
    return n.value


def render_fwd(cfg, packed_fwd0, packed_fwd1, origins, directions, viewdirs, randomized=False, t_rand=None, u=None,
               seed=0, ws=None):
    """NerfModel.__call__ forward: [(rgb,disp,acc)_coarse, (rgb,disp,acc)_fine]."""
    _require_gpu()
    lib = _lib.load()
    B = origins.shape[0]
    dev = origins.device
    nbytes = render_workspace_bytes(cfg, B)
    if ws is None or ws.numel() < nbytes:
        ws = _new(max(nbytes, 16), device=dev, dtype=torch.uint8)
    outs = [(_new(B, 3, device=dev), _new(B, device=dev), _new(B, device=dev))]
    fine = cfg.num_fine_samples > 0
    if fine:
        outs.append((_new(B, 3, device=dev), _new(B, device=dev), _new(B, device=dev)))
    f = outs[1] if fine else (None, None, None)
    check(lib.pxo_render_fwd(ctypes.byref(cfg), _f(packed_fwd0), _f(packed_fwd1), _f(origins), _f(directions),
                             _f(viewdirs), B, int(randomized), _f(t_rand), _f(u), seed, _f(outs[0][0]),
                             _f(outs[0][1]), _f(outs[0][2]), _f(f[0]), _f(f[1]), _f(f[2]), _p(ws), ws.numel(),
                             _stream()), "pxo_render_fwd")
    return outs


def train_fwd_bwd(cfg, params, packed, origins, directions, viewdirs, pixels, grads, stats, ws, randomized=True,
                  t_rand=None, u=None, sp_points=None, seed=0, grads0_ready=None):
    """loss_fn + value_and_grad on this device's shard; fills grads (2-MLP arena) and stats[6].
    grads0_ready: an `Event` recorded on the current stream once MLP_0's half of `grads` is final (the coarse level is
    reversed before the fine level starts), or None."""
    _require_gpu()
    lib = _lib.load()
    (f0, b0), (f1, b1) = packed
    B = origins.shape[0]
    check(lib.pxo_train_fwd_bwd_bucketed(ctypes.byref(cfg), _f(params), _f(f0), _f(b0), _f(f1), _f(b1), _f(origins),
                                         _f(directions), _f(viewdirs), _f(pixels), B, int(randomized), _f(t_rand), _f(u),
                                         _f(sp_points), seed, _f(grads), _f(stats), _p(ws), ws.numel(),
                                         grads0_ready.handle if grads0_ready is not None else None, _stream()),
          "pxo_train_fwd_bwd_bucketed")


def train_backward_work(cfg, B, ws):
    """(live, total) 16-row chunks of the reverse pass of the last train_fwd_bwd call on workspace `ws` (synchronises)."""
    _require_gpu()
    live, total = ctypes.c_int64(0), ctypes.c_int64(0)
    check(_lib.load().pxo_train_backward_work(ctypes.byref(cfg), B, _p(ws), ws.numel(), ctypes.byref(live), ctypes.byref(total),
                                              _stream()), "pxo_train_backward_work")
    return live.value, total.value


class Event:
    """hipEvent_t owned through the C ABI (pxo_event_create): recorded by pxo_train_fwd_bwd_bucketed, waited for by
    `wait(stream)`."""

    def __init__(self):
        _require_gpu()
        h = ctypes.c_void_p(None)
        check(_lib.load().pxo_event_create(ctypes.byref(h)), "pxo_event_create")
        self.handle = h

    def wait(self, stream):
        """`stream` (a torch.cuda.Stream) continues once the last record of this event has completed."""
        check(_lib.load().pxo_stream_wait_event(ctypes.c_void_p(stream.cuda_stream), self.handle), "pxo_stream_wait_event")

    def __del__(self):
        try:
            if self.handle:
                _lib.load().pxo_event_destroy(self.handle)
        except Exception:
            pass


def eval_points(cfg, packed_fwd, points, want_rgb=True):
    """NerfModel.eval_points_raw: raw_rgb [N,3K] (or None), raw_sigma [N,1]."""
    _require_gpu()
    lib = _lib.load()
    points = points.reshape(-1, 3)
    N = points.shape[0]
    dev = points.device
    raw_rgb = _new(N, rgb_channels(cfg), device=dev) if want_rgb else None
    raw_sigma = _new(N, 1, device=dev)
    check(lib.pxo_eval_points(ctypes.byref(cfg), _f(packed_fwd), _f(points), N, _f(raw_rgb), _f(raw_sigma), _stream()),
          "pxo_eval_points")
    return raw_rgb, raw_sigma


def grid_sigma(cfg, packed_fwd, reso, x0, x1, offset, scale, out=None):
    """sigma on the dense grid slab x in [x0,x1) (octree/extraction.py:290-320)."""
    _require_gpu()
    lib = _lib.load()
    n = (x1 - x0) * reso * reso
    if out is None:
        out = _new(n, device=packed_fwd.device)
    off = (ctypes.c_float * 3)(*[float(v) for v in offset])
    sc = (ctypes.c_float * 3)(*[float(v) for v in scale])
    check(lib.pxo_grid_sigma(ctypes.byref(cfg), _f(packed_fwd), reso, x0, x1, off, sc, _f(out), _stream()),
          "pxo_grid_sigma")
    return out


PROF_MLP_FWD, PROF_MLP_BWD_DATA, PROF_WGRAD_MAIN, PROF_WGRAD_OTHER, PROF_NUM_TAGS = 0, 1, 2, 3, 4


TUNE_TILE_SCHED, TUNE_WGRAD_RANGES, TUNE_WGRAD_SKINNY_RANGES, TUNE_COARSE_REVERSE_STREAM, TUNE_X6_WGRAD = 0, 1, 2, 3, 4        # PXO_TUNE_* of include/plenoctree_hip.h


def set_tuning(knob, value):
    """pxo_set_tuning: choose between implementations of the same result (A/B sessions, equality tests)."""
    check(_lib.load().pxo_set_tuning(int(knob), int(value)), "pxo_set_tuning")


def get_tuning(knob):
    v = ctypes.c_int(0)
    check(_lib.load().pxo_get_tuning(int(knob), ctypes.byref(v)), "pxo_get_tuning")
    return v.value


def occupy_cus(blocks, threads, micros, stream=None, lds_bytes=0):
    """pxo_occupy_cus: `blocks` idle workgroups (each with `lds_bytes` of LDS) for `micros` us on `stream` (a torch.cuda.Stream;
    default: the current one)."""
    _require_gpu()
    h = ctypes.c_void_p(stream.cuda_stream) if stream is not None else _stream()
    check(_lib.load().pxo_occupy_cus(int(blocks), int(threads), float(micros), int(lds_bytes), h), "pxo_occupy_cus")


def profile_enable(on=True, tags=None):
    """HIP-event brackets around the tagged kernel launches: all tags (on=True), none (False), or the listed `tags`.
    Every bracket costs the stream two event records (~5 us each between kernels that would otherwise run back to back)."""
    mask = 0
    if tags is not None:
        for t in tags:
            mask |= 1 << int(t)
    elif on:
        mask = (1 << PROF_NUM_TAGS) - 1
    check(_lib.load().pxo_profile_enable(mask), "pxo_profile_enable")


def profile_read(tag):
    """(launches, total_ms, total_rows) of the kernels tagged `tag` since the last read."""
    n, ms, rows = ctypes.c_int64(0), ctypes.c_double(0.0), ctypes.c_int64(0)
    check(_lib.load().pxo_profile_read(tag, ctypes.byref(n), ctypes.byref(ms), ctypes.byref(rows)), "pxo_profile_read")
    return n.value, ms.value, rows.value



def randint(seed, stream_id, count, n, device=None):
    """count uniform integers in [0, n) (int64) from the Philox stream."""
    _require_gpu()
    out = _new(count, device=device, dtype=torch.int64)
    check(_lib.load().pxo_randint(seed, stream_id, count, n, _p(out), _stream()), "pxo_randint")
    return out


def generate_rays(c2w, W, H, focal, pixel_ids=None, count=None):
    """Rays of the given pixels (or of pixels 0..count-1) of one camera: (origins, directions, viewdirs)."""
    _require_gpu()
    c2w = c2w[:3, :4].contiguous()
    B = pixel_ids.shape[0] if pixel_ids is not None else (count if count is not None else W * H)
    dev = c2w.device
    o, d, v = _new(B, 3, device=dev), _new(B, 3, device=dev), _new(B, 3, device=dev)
    if pixel_ids is not None and pixel_ids.dtype != torch.int64:
        raise PxoError("pixel_ids must be int64")
    check(_lib.load().pxo_generate_rays(_f(c2w), W, H, float(focal), _p(pixel_ids), B, _f(o), _f(d), _f(v), _stream()),
          "pxo_generate_rays")
    return o, d, v


def sample_batch(seed, stream_id, c2w, W, H, focal, image_rgb, B, want_ids=False, first=0):
    """One training batch of one image in one launch: (origins, directions, viewdirs, pixels[, pixel_ids]); `first`: the B rows
    are elements first .. first + B - 1 of the stream (a rank's shard of one global draw)."""
    _require_gpu()
    c2w = c2w[:3, :4].contiguous()
    dev = c2w.device
    o, d, v, px = _new(B, 3, device=dev), _new(B, 3, device=dev), _new(B, 3, device=dev), _new(B, 3, device=dev)
    ids = _new(B, device=dev, dtype=torch.int64) if want_ids else None
    check(_lib.load().pxo_sample_batch(seed, stream_id, _f(c2w), W, H, float(focal), _f(image_rgb), B, int(first), _p(ids), _f(o), _f(d),
                                       _f(v), _f(px), _stream()), "pxo_sample_batch")
    return (o, d, v, px, ids) if want_ids else (o, d, v, px)


def generate_rays_multi(c2w_all, W, H, focal, ray_ids):
    """Rays of ids into the flattened [n_cams, H*W] table (image_batching): (origins, directions, viewdirs)."""
    _require_gpu()
    if ray_ids.dtype != torch.int64:
        raise PxoError("ray_ids must be int64")
    c2w_all = c2w_all[:, :3, :4].contiguous()
    B = ray_ids.shape[0]
    dev = c2w_all.device
    o, d, v = _new(B, 3, device=dev), _new(B, 3, device=dev), _new(B, 3, device=dev)
    check(_lib.load().pxo_generate_rays_multi(_f(c2w_all), c2w_all.shape[0], W, H, float(focal), _p(ray_ids), B,
                                              _f(o), _f(d), _f(v), _stream()), "pxo_generate_rays_multi")
    return o, d, v


def mean_over_samples(cfg, raw_rgb, raw_sigma, samples_per_cell, out=None):
    _require_gpu()
    n = raw_sigma.numel() // samples_per_cell
    if out is None:
        out = _new(n, rgb_channels(cfg) + 1, device=raw_sigma.device)
    elif out.numel() != n * (rgb_channels(cfg) + 1):
        raise PxoError("mean_over_samples: `out` has the wrong size")
    check(_lib.load().pxo_mean_over_samples(ctypes.byref(cfg), _f(raw_rgb), _f(raw_sigma.reshape(-1)), n,
                                            samples_per_cell, _f(out), _stream()), "pxo_mean_over_samples")
    return out
