"""Oracle parity in the TRAINED regime (-m gpu).

north_star states its bars "on chair", i.e. for a converged model (the reference measures PSNR on the trained network,
nerf_sh/train.py:245-268).  The BASELINE-size tests of tests/test_gpu_fullsize.py draw random weights (a fuzzy volume,
PSNR ~ 10 dB); here the network is first TRAINED by the HIP path -- 2,000 steps of 4,096 rays on datasets.Synthetic from
the fixed-seed initialisation, the `converge` leg of bench.py, ~28 dB on held-out views -- and the comparisons are made at
that state: sharp surfaces, most sample rows exactly dead (relu(sigma) = 0 or transmittance underflow), sharply peaked
inverse-CDF sampling.

  test_trained_step_matches_f64_oracle   ONE full 4,096-ray step (+ 10k sparsity points) with injected t_rand / u / sp_points
                                         through pxo_train_fwd_bwd, dense AND skip_zero_rows = 1, against loss_fn +
                                         value_and_grad of the oracle (nerf_sh/train.py:68-116) in float64
  test_trained_render_matches_f64_oracle every 4th pixel (both axes) of one held-out 800 x 800 view at those weights, HIP vs
                                         the float64 oracle's render of the SAME weights: |dPSNR| <= 1e-4 dB
  test_trained_psnr_twin_long            HIP-trained vs ORACLE-trained (the oracle's 1,500 steps of 1,024 rays are a fixture):
                                         held-out PSNR within 0.1 dB at ~24 dB

Bounds, fixed numbers:
  Stats      against the float64 oracle: rtol 2e-5, or within STATS_FACTOR x the distance of the oracle's own float32
             evaluation from its float64 one (with random u the fine samples of a float32 and a float64 evaluation differ in
             the bins next to empty stretches of the cdf: the fine loss of this batch moves by 3e-5 (oracle float32) / 3.8e-5 (HIP) relative, r05d);
             loss_sp 5e-3 (1 - mean(exp(-0.05 relu(sigma))) of a mostly empty volume is a difference of nearly equal numbers
             in float32; same bar as L1 in tests/test_gpu_reference_fixtures.py)
  gradient   relative L2 error per MLP against the float64 oracle <= GRAD_FACTOR x the distance of the oracle's OWN float32
             evaluation from its float64 one (the bound G1 uses: the float32 noise floor of this function at this state),
             and never looser than GRAD_CAP
  skipping   gradients and Stats bit-identical to the dense pass; live 16-row chunk fraction recorded
"""
import json
import os
import time

import pytest
import torch

from oracle import nerf_oracle as O
from _helpers import _gpu, _ops, _psnr, oracle_loss_and_grad_chunked, oracle_render_chunked

pytestmark = pytest.mark.gpu

TRAIN_STEPS, RAYS = 2000, 4096
GRAD_FACTOR, GRAD_CAP = 2.0, 1e-2
STATS_FACTOR = 2.0


def _record(name, **kv):
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "trained_state_parity.jsonl"), "a") as f:
            f.write(json.dumps(dict(test=name, **kv)) + "\n")


@pytest.fixture(scope="module")
def trained():
    """2,000 HIP train steps of 4,096 rays (the blender preset, Philox draws, the reference's lr schedule) on the analytic
    800 x 800 scene.  The warm start runs with zero-row skipping (bit-identical gradients, twice as fast on this scene)."""
    _ops(); dev = _gpu()
    from plenoctree_amd.nerf_sh.nerf import datasets, models, utils
    args = utils.define_flags().parse_args([])
    args.config = "blender"; utils.update_flags(args)
    args.dataset = "synthetic"; args.batch_size = RAYS; args.factor = 0; args.train_dir = "/tmp/pxo_trained_state"
    args.skip_zero_rows = True
    model, params = models.construct_nerf(args, dev)
    state = models.TrainState(model.cfg, params)
    ds = datasets.Synthetic("train", args, dev, batch_size=RAYS)
    t0 = time.time()
    for step in range(TRAIN_STEPS):
        lr = utils.learning_rate_decay(step, args.lr_init, args.lr_final, args.max_steps)
        models.train_step(model, state, next(ds), lr, randomized=True, seed=step << 8)
    torch.cuda.synchronize()
    stats = dict(zip(utils.Stats._fields, state.stats.cpu().tolist()))
    _record("warm_start", steps=TRAIN_STEPS, rays=RAYS, train_s=time.time() - t0, last_batch_psnr=stats["psnr"])
    assert stats["psnr"] > 22.0, f"warm start did not leave the untrained regime: {stats}"
    return dict(args=args, model=model, state=state, ds=ds, dev=dev)


_ORACLE = {}


@pytest.mark.parametrize("prec_name,prec", [("f32", 0), ("bf16x6", 2)])
def test_trained_step_matches_f64_oracle(trained, prec_name, prec):
    """HIP legs in native float32 and in the opt-in bf16x6 emulation (csrc/mlp_x6_kernels.hip), same batch (the fixture's
    sampler is re-seeded per leg through the module-level cache of the batch), same bounds."""
    ops = _ops(); dev = trained["dev"]
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    args, state = trained["args"], trained["state"]
    cfg = O.Cfg()                       # blender preset = the oracle's defaults (asserted below)
    assert (cfg.sh_deg, cfg.near, cfg.far, cfg.sparsity_npoints) == (args.sh_deg, args.near, args.far, args.sparsity_npoints)
    if "batch" not in _ORACLE:
        _ORACLE["batch"] = next(trained["ds"])          # 4,096 random pixels of one training image (one batch for both legs)
    batch = _ORACLE["batch"]
    rays_dev = batch["rays"]; px_dev = batch["pixels"]
    gen = torch.Generator().manual_seed(20200823)
    t_rand = torch.rand(RAYS, 64, generator=gen); u = torch.rand(RAYS, 128, generator=gen)
    sp_pts = (torch.rand(cfg.sparsity_npoints, 3, generator=gen) * 2 - 1) * cfg.sparsity_radius
    fd = state.params.clone()
    n = fd.numel() // 2
    outs = {}
    for skip in (0, 1):
        pcfg = type(trained["model"].cfg).from_buffer_copy(trained["model"].cfg)      # PxoCfg is a ctypes struct
        pcfg.skip_zero_rows = skip
        pcfg.mlp_precision = prec
        packed = [ops.pack_weights(pcfg, fd[i * n:(i + 1) * n].contiguous()) for i in range(2)]
        grads = torch.full_like(fd, float("nan")); stats = torch.zeros(6, device=dev)
        ws = torch.empty(ops.train_workspace_bytes(pcfg, RAYS), dtype=torch.uint8, device=dev)
        ws.fill_(0xFF)
        ops.train_fwd_bwd(pcfg, fd, packed, rays_dev.origins, rays_dev.directions, rays_dev.viewdirs, px_dev, grads, stats, ws,
                          randomized=True, t_rand=t_rand.to(dev), u=u.to(dev), sp_points=sp_pts.to(dev))
        live, total = ops.train_backward_work(pcfg, RAYS, ws)
        outs[skip] = (grads.cpu(), stats.cpu(), live, total)
        del ws
    (g0, s0, l0, t0_), (g1, s1, l1, t1_) = outs[0], outs[1]
    assert bool(torch.isfinite(g0).all()) and bool(torch.isfinite(g1).all())
    assert torch.equal(g0, g1) and torch.equal(s0, s1), "skip_zero_rows changed bits at the trained state"
    assert l0 == t0_ == t1_ and 0 < l1 < 0.5 * t1_, (l0, t0_, l1, t1_)      # a trained scene: most chunks are exactly dead

    flat = fd.cpu()
    rays = O.Rays(*[r.cpu() for r in rays_dev]); px = px_dev.cpu()
    tc = time.time()
    if "oracle" not in _ORACLE:
        _ORACLE["oracle"] = (oracle_loss_and_grad_chunked(flat, rays, px, cfg, t_rand, u, sp_pts, torch.float32),
                             oracle_loss_and_grad_chunked(flat, rays, px, cfg, t_rand, u, sp_pts, torch.float64))
    (st32, g32), (st64, g64) = _ORACLE["oracle"]
    t_cpu = time.time() - tc
    rec = dict(live_chunk_fraction=l1 / t1_, oracle_s=t_cpu, psnr_batch_f64=st64["psnr"], psnr_batch_f32=st32["psnr"],
               psnr_batch_hip=float(s0[1]))
    for i, k in enumerate(("loss", "psnr", "loss_c", "loss_sp", "psnr_c", "weight_l2")):
        rec[f"stats_{k}_rel"] = abs(float(s0[i]) - st64[k]) / max(abs(st64[k]), 1e-30)
        rec[f"stats_{k}_rel_f32_oracle"] = abs(st32[k] - st64[k]) / max(abs(st64[k]), 1e-30)
    g_hip = g0.double()
    errs = []
    for mi, (lo, hi) in enumerate(((0, n), (n, 2 * n))):
        ref = g64[lo:hi]
        assert float(ref.norm()) > 1e-6, "degenerate test: oracle gradient vanishes"
        e_hip = float((g_hip[lo:hi] - ref).norm() / ref.norm())
        e_cpu = float((g32[lo:hi].double() - ref).norm() / ref.norm())
        rec[f"mlp{mi}_hip"] = e_hip; rec[f"mlp{mi}_f32_oracle"] = e_cpu; rec[f"mlp{mi}_grad_norm"] = float(ref.norm())
        errs.append((mi, e_hip, e_cpu))
    _record(f"trained_step[{prec_name}]", **rec)
    print("trained step:", json.dumps(rec))
    for i, k in enumerate(("loss", "psnr", "loss_c", "loss_sp", "psnr_c", "weight_l2")):
        rtol = 5e-3 if k == "loss_sp" else 2e-5
        tol = max(rtol * abs(st64[k]), STATS_FACTOR * abs(st32[k] - st64[k]))
        assert abs(float(s0[i]) - st64[k]) <= tol, (k, float(s0[i]), st64[k], st32[k], tol)
    for mi, e_hip, e_cpu in errs:
        bound = min(max(GRAD_FACTOR * e_cpu, 1e-3), GRAD_CAP)
        assert e_hip <= bound, f"MLP_{mi}: rel L2 err vs f64 oracle {e_hip:.3g} > {bound:.3g} (CPU f32 oracle: {e_cpu:.3g})"


@pytest.mark.parametrize("prec_name,prec", [("f32", 0), ("bf16x6", 2)])
def test_trained_render_matches_f64_oracle(trained, prec_name, prec):
    """render_image's per-chunk call (pxo_render_fwd, deterministic sampling as nerf_sh/eval.py:57) on every 4th pixel in
    both axes of one held-out 800 x 800 view (40,000 rays) at the trained weights, against the float64 oracle's render of
    the same weights; PSNRs are taken against the view's ground truth as nerf_sh/train.py:245-268 does."""
    _ops(); dev = trained["dev"]
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    from plenoctree_amd.nerf_sh.nerf import datasets, utils
    model, state = trained["model"], trained["state"]
    if prec:        # the same weights through the opt-in bf16x6 kernels (their own weight images)
        from plenoctree_amd.nerf_sh.nerf import models
        cfg6 = type(model.cfg).from_buffer_copy(model.cfg); cfg6.mlp_precision = prec
        model, state = models.NerfModel(cfg6), models.TrainState(cfg6, state.params.clone())
    test = datasets.Synthetic("test", trained["args"], dev)
    ex = test.get_image(67)
    sub = lambda t: t[::4, ::4].reshape(-1, 3).contiguous()
    rays_dev = utils.Rays(*[sub(r) for r in ex["rays"]]); px = sub(ex["pixels"]).cpu()
    outs = []
    for i0 in range(0, px.shape[0], 8192):
        r = utils.Rays(*[t[i0:i0 + 8192].contiguous() for t in rays_dev])
        outs.append(model.apply(state, r, False))
    rgb_hip = [torch.cat([o[lvl][0] for o in outs]).cpu() for lvl in range(2)]
    acc_hip = torch.cat([o[1][2] for o in outs]).cpu()
    cfg = O.Cfg()
    rays = O.Rays(*[r.cpu() for r in rays_dev])
    tc = time.time()
    if "render64" not in _ORACLE:
        _ORACLE["render64"] = oracle_render_chunked(state.params.cpu(), rays, cfg, None, None, torch.float64, chunk=2048)
    ref64 = _ORACLE["render64"]
    t_cpu = time.time() - tc
    p_hip, p64 = _psnr(rgb_hip[1], px), _psnr(ref64[1][0], px)
    p_hip_c, p64_c = _psnr(rgb_hip[0], px), _psnr(ref64[0][0], px)
    err = (rgb_hip[1].double() - ref64[1][0]).abs().max(dim=-1)[0]
    rec = dict(rays=int(px.shape[0]), psnr_hip=p_hip, psnr_f64=p64, d_psnr=abs(p_hip - p64), psnr_hip_coarse=p_hip_c,
               psnr_f64_coarse=p64_c, d_psnr_coarse=abs(p_hip_c - p64_c), image_psnr_hip_vs_f64=_psnr(rgb_hip[1], ref64[1][0]),
               max_err=float(err.max()), n_over_1e3=int((err > 1e-3).sum()), median_err=float(err.median()),
               acc_max_err=float((acc_hip.double() - ref64[1][2]).abs().max()),
               background_fraction=float((ref64[1][2] < 1e-3).double().mean()), oracle_s=t_cpu)
    _record(f"trained_render[{prec_name}]", **rec)
    print("trained render:", json.dumps(rec))
    assert p64 > 22.0, p64                                   # the trained regime, not the 14 dB one
    assert abs(p_hip - p64) <= 1e-4, (p_hip, p64)
    assert abs(p_hip_c - p64_c) <= 1e-4, (p_hip_c, p64_c)
    # the image itself: no pixel far off, and at most a handful beyond 1e-3 (fine samples next to an empty stretch of the cdf)
    assert float(err.max()) <= 2e-2 and int((err > 1e-3).sum()) <= 40, (float(err.max()), int((err > 1e-3).sum()))


def test_trained_psnr_twin_long(golden_dir):
    """The 0.1 dB bar of north_star OUTSIDE the 14 dB regime: 1,500 Adam steps of 1,024 rays x (64+128) samples + 10,000 sparsity
    points with the reference's lr schedule annealed over the horizon -- held-out PSNR ~24 dB from 10 dB.  The oracle leg
    (float32, 1.6 CPU-hours) was run once by `tests/golden/make_trained_twin.py long`; its final parameters are the fixture
    trained_twin_1024x1500.npz.  This test replays the same batches and injected randoms (tests/_helpers.py:twin_steps -- seeds
    only) through the HIP path and compares held-out PSNRs (every 4th pixel of three test views, deterministic sampling).

    What "the PSNR of this run" means had to be measured first (scripts/twin_noise_sensitivity.py,
    profiles/r05h_twin_noise_sensitivity.jsonl): the 1,500-step trajectory is chaotic -- multiplying every step's gradient
    element-wise by (1 + 1e-6 N(0,1)), i.e. the size of a re-ordered float32 sum, moves the final held-out PSNR of the HIP run
    anywhere in 23.93 .. 24.01 dB (1e-4 and 1e-3: 23.95 .. 24.02 dB; ten legs: mean 23.968, sd 0.031, range 0.093 dB).  A single
    pair of float32 runs is therefore only defined to ~ +-0.05 dB, and the unperturbed HIP leg happens to sit at the bottom of
    that cloud (23.931) while the float32 oracle's run (24.043; 24.053 / 24.050 on 7 / 5 threads) sits just above its top: 0.112 dB
    apart as a pair.  The FLOAT64 oracle's run of the same steps ends at 23.959 dB: 0.028 dB from the unperturbed HIP leg, in the
    middle of the HIP cloud, 0.09 dB below the float32 oracle -- the float32 oracle is the outlier, not HIP.  The test compares
    (i) the HIP leg with the float64 oracle's value as a single pair and (ii) the float32 oracle's value with the MEAN of four
    HIP legs (unperturbed + three 1e-6-perturbed ones, 10 s of GPU each):
      |PSNR(HIP-trained, HIP-rendered) - PSNR(float64-oracle-trained)| <= 0.1 dB                     (north_star, single pair)
      PSNR(oracle-trained, oracle-rendered) >= 22 dB                                                (the regime)
      |mean PSNR(HIP legs) - PSNR(oracle-trained, oracle-rendered)| <= 0.1 dB                        (north_star)
      every HIP leg within 0.15 dB of the oracle, the legs within 0.12 dB of each other              (the cloud, measured 0.08)
      |PSNR(HIP-trained, HIP-rendered) - PSNR(HIP-trained, float64-oracle-rendered)| <= 1e-4 dB     (same weights, unperturbed leg)
    Every number, including the single-pair distance, goes to gpurun_out/trained_state_parity.jsonl."""
    ops = _ops(); dev = _gpu()
    import numpy as np
    from _helpers import TWIN_LONG_RAYS, TWIN_LONG_STEPS, pxo_cfg, twin_heldout, twin_steps
    from plenoctree_amd.nerf_sh.nerf import models, utils
    path = os.path.join(golden_dir, f"trained_twin_{TWIN_LONG_RAYS}x{TWIN_LONG_STEPS}.npz")
    if not os.path.exists(path):
        pytest.skip(f"{path} missing: the oracle leg is 1.6 CPU-hours, `python tests/golden/make_trained_twin.py long`")
    g = np.load(path)
    assert int(g["rays_per_step"]) == TWIN_LONG_RAYS and int(g["steps"]) == TWIN_LONG_STEPS
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    cfg = O.Cfg()
    pcfg = pxo_cfg(ops, cfg)
    flat0 = O.flatten_params(O.init_params(cfg, seed=20200823))
    model = models.NerfModel(pcfg)
    rays, px = twin_heldout()
    drays = utils.Rays(*[r.to(dev) for r in rays])
    B = TWIN_LONG_RAYS
    feed = [(utils.Rays(*[r.to(dev) for r in b["rays"]]), b["pixels"].to(dev), t.to(dev), u.to(dev), sp.to(dev), lr)
            for _, b, t, u, sp, lr in twin_steps(B, TWIN_LONG_STEPS, cfg)]

    def hip_leg(eps, seed):
        """train_step (ops.train_fwd_bwd + ops.adam_pack_step, as models.train_step sequences them) with the gradient of every
        step multiplied by (1 + eps N(0,1)); eps = 0: the plain product path."""
        state = models.TrainState(pcfg, flat0.clone().to(dev))
        gen = torch.Generator(device=dev).manual_seed(1000 + seed)
        ws = state.workspace(ops.train_workspace_bytes(pcfg, B))
        for r, pixels, t_rand, u, sp, lr in feed:
            ops.train_fwd_bwd(pcfg, state.params, state.packed, r.origins, r.directions, r.viewdirs, pixels, state.grads,
                              state.stats, ws, randomized=True, t_rand=t_rand, u=u, sp_points=sp)
            if eps > 0:
                state.grads.mul_(1.0 + eps * torch.randn(state.grads.shape, device=dev, generator=gen))
            ops.adam_pack_step(pcfg, state.params, state.m, state.v, state.grads, lr, state.step, state.packed)
            state.step += 1
        return state, _psnr(model.apply(state, drays, False)[1][0].cpu(), px)

    state, psnr_hip = hip_leg(0.0, 0)
    legs = [psnr_hip] + [hip_leg(1e-6, seed)[1] for seed in (1, 2, 3)]
    with torch.no_grad():
        ref = O.render(O.unflatten_params(torch.tensor(g["params"]), cfg), rays, cfg)[1][0]
        rays64 = O.Rays(*[r.double() for r in rays])
        cross = O.render(O.unflatten_params(state.params.cpu().double(), cfg), rays64, cfg)[1][0]
    psnr_ref, psnr_cross = _psnr(ref, px), _psnr(cross, px)
    mean_hip = sum(legs) / len(legs)
    rec = dict(steps=TWIN_LONG_STEPS, rays_per_step=TWIN_LONG_RAYS, psnr_init=float(g["psnr_init"]), psnr_oracle_trained=psnr_ref,
               psnr_hip_trained=psnr_hip, psnr_hip_trained_f64_oracle_rendered=psnr_cross, psnr_hip_legs=legs,
               psnr_hip_legs_mean=mean_hip, psnr_hip_legs_range=max(legs) - min(legs), d_single_pair=abs(psnr_hip - psnr_ref),
               d_mean_vs_oracle=abs(mean_hip - psnr_ref))
    _record("trained_twin_long", **rec)
    print("long twin:", json.dumps(rec))
    # the float64 oracle's run of the same 1,500 steps (`make_trained_twin.py long 8 f64`, 2.6 CPU-hours; PSNR only, no weights):
    # the arbiter, as everywhere else in this suite.  It ends at 23.959 dB -- in the middle of the HIP cloud and 0.09 dB below
    # the float32 oracle's three runs (24.043 / 24.053 / 24.050): the float32 ORACLE is the high one, not HIP.
    f64_path = os.path.join(golden_dir, f"trained_twin_{TWIN_LONG_RAYS}x{TWIN_LONG_STEPS}_threads8_f64.json")
    if os.path.exists(f64_path):
        with open(f64_path) as f:
            psnr_f64 = float(json.load(f)["psnr_trained"])
        rec["psnr_f64_oracle_trained"] = psnr_f64; rec["d_single_pair_vs_f64"] = abs(psnr_hip - psnr_f64)
        _record("trained_twin_long_vs_f64", psnr_f64_oracle_trained=psnr_f64, psnr_hip_trained=psnr_hip,
                d_single_pair_vs_f64=abs(psnr_hip - psnr_f64), d_mean_vs_f64=abs(mean_hip - psnr_f64))
        assert abs(psnr_hip - psnr_f64) <= 0.1, (psnr_hip, psnr_f64)           # north_star as a SINGLE pair, against float64
        assert abs(mean_hip - psnr_f64) <= 0.1, (mean_hip, psnr_f64)
    assert psnr_ref == pytest.approx(float(g["psnr_trained"]), abs=2e-3)        # the fixture's weights render as recorded
    assert psnr_ref >= 22.0 and psnr_ref > float(g["psnr_init"]) + 10.0          # left the 14 dB regime
    assert abs(mean_hip - psnr_ref) <= 0.1, (mean_hip, psnr_ref, legs)
    assert max(abs(x - psnr_ref) for x in legs) <= 0.15 and max(legs) - min(legs) <= 0.12, (legs, psnr_ref)
    assert abs(psnr_hip - psnr_cross) <= 1e-4, (psnr_hip, psnr_cross)
    _ORACLE["twin_long"] = (psnr_hip, psnr_ref)


@pytest.mark.xfail(reason="KNOWN, kept visible: as a SINGLE pair against the FLOAT32 oracle's run the long twin is 0.11 dB apart "
                          "(23.93 vs 24.04 dB), above north_star's 0.1 dB.  The bar is met against the float64 oracle (0.03 dB, "
                          "single pair) and by the mean of four HIP legs (0.08 dB vs float32, 0.002 dB vs float64): "
                          "test_trained_psnr_twin_long, DESIGN.md section 2.", strict=False)
def test_trained_psnr_twin_long_single_pair_vs_f32_oracle():
    """The round-5 advisor's item: the original single-pair criterion stays in the suite as an expected failure, so that the
    0.11 dB is a number every run prints, not a sentence in a document."""
    if "twin_long" not in _ORACLE:
        pytest.skip("test_trained_psnr_twin_long did not run in this session")
    psnr_hip, psnr_f32_oracle = _ORACLE["twin_long"]
    assert abs(psnr_hip - psnr_f32_oracle) <= 0.1, (psnr_hip, psnr_f32_oracle)
