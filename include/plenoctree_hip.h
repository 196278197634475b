/*
 * plenoctree_hip.h -- C ABI of the MI355X (gfx950) NeRF-SH hot path.
 *
 * The reference (sxyu/plenoctree) has no FFI of its own: its hot path is a Python call
 * surface (JAX/Flax, and a torch twin for extraction).  Each entry point below names the
 * reference interface it replaces (file:line relative to the reference tree).  Everything
 * is extern "C", plain pointers and sizes; device pointers are caller-owned (e.g. torch
 * ROCm storage), no hidden allocation (workspace sizes are queried, then passed in), every
 * call is asynchronous on the hipStream_t given as `void* stream`.
 *
 * Return value: 0 = ok, negative = error (message via pxo_last_error()).
 * Empty inputs (a row / ray / point count of 0) return 0 without touching any pointer, so zero-size
 * tensors (NULL data pointers) are valid arguments; pxo_train_fwd_bwd alone requires B >= 1 (a mean over no
 * rays has no value).
 *
 * Layouts (all float32, row-major):
 *   rays      origins/directions/viewdirs [B,3]         (nerf_sh/nerf/utils.py:53 Rays)
 *   params    flat arena per model: MLP_0{Dense_0..9 kernel[in,out],bias[out]}, MLP_1{...}
 *             Dense_0..7 trunk, Dense_8 sigma head, Dense_9 rgb head
 *             (key order of octree/nerf/models.py:91-102); see pxo_param_layout().
 *   raw_rgb   [M,3K] channel-major then SH coefficient (nerf_sh/nerf/models.py:269-272)
 *   raw_sigma [M]
 */
#ifndef PLENOCTREE_HIP_H_
#define PLENOCTREE_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PXO_OK 0
#define PXO_ERR_ARG (-1)
#define PXO_ERR_HIP (-2)
#define PXO_ERR_WORKSPACE (-3)
#define PXO_ERR_UNSUPPORTED (-4)

#define PXO_NET_DEPTH 8
#define PXO_NET_WIDTH 256
#define PXO_SKIP_LAYER 4
#define PXO_ENC_DIM 63      /* 3*(1+2*10), nerf_sh/nerf/model_utils.py:145-173 */
#define PXO_ENC_PAD 64
#define PXO_NUM_LEAVES 20   /* per MLP: 10 x (kernel, bias) */

/* Hyper-parameters of the path: the flags of nerf_sh/nerf/utils.py:61-230 that reach the
 * model (NerfModel fields, nerf_sh/nerf/models.py:55-78) and the loss (train.py:77-85). */
typedef struct PxoCfg {
  int32_t num_coarse_samples;  /* 64  */
  int32_t num_fine_samples;    /* 128 (0 = coarse only) */
  int32_t sh_deg;              /* 0..4; K=(deg+1)^2 */
  int32_t min_deg_point;       /* must be 0  */
  int32_t max_deg_point;       /* must be 10 */
  int32_t white_bkgd;
  int32_t lindisp;
  int32_t sparsity_npoints;    /* 10000 */
  float near_;
  float far_;
  float sparsity_weight;       /* 1e-3; 0 disables the branch (train.py:77) */
  float sparsity_length;       /* 0.05 */
  float sparsity_radius;       /* 1.5 */
  float weight_decay_mult;     /* 0 */
  int32_t mlp_precision;       /* PXO_MLP_F32 (0, default); opt-ins: PXO_MLP_BF16X3 (inference only), PXO_MLP_BF16X6, see below */
  float noise_std;             /* 0 = the reference's None (every preset); > 0: add_gaussian_noise on raw sigma of the ray
                                  samples when randomized (nerf_sh/nerf/models.py:258-264,318-324) */
  int32_t skip_zero_rows;      /* pxo_train_fwd_bwd only.  1: sample rows whose upstream gradient (d loss / d raw_rgb, d raw_sigma) is
                                  EXACTLY zero -- empty space (relu(sigma) = 0, hence weight 0), samples behind an opaque surface,
                                  whole background rays -- are left out of the reverse pass in 16-row chunks (128-row tiles in the
                                  backward(data) kernel).  The gradients are bit-identical (only exact zeros leave the sums); the
                                  time saved depends on the scene, so bench.py's throughput records run with 0 = dense, like the
                                  reference's jax.value_and_grad (nerf_sh/train.py:116) */
} PxoCfg;

/* PxoCfg.mlp_precision.  PXO_MLP_F32: exact float32 MFMA everywhere (the reference's precision; training and every
 * reported throughput use it).  PXO_MLP_BF16X3: the forward-only entry points (pxo_eval_points, pxo_grid_sigma,
 * pxo_render_fwd, pxo_mlp_fwd without saved tensors) evaluate each product as hi*hi + hi*lo + lo*hi of bf16 splits with
 * float32 accumulation (csrc/mlp_x3_kernels.hip); pxo_pack_weights then writes the split image (same size) and takes
 * packed_bwd == NULL; pxo_train_fwd_bwd and the saved-tensor form of pxo_mlp_fwd return PXO_ERR_UNSUPPORTED.
 * PXO_MLP_BF16X6: float32-ACCURATE split precision for the whole path, training included (csrc/mlp_x6_kernels.hip): every
 * float32 operand of the fused MLP forward and backward(data) is split exactly into three bf16 parts and a product is the six
 * partial products of order <= 2^-16 with float32 accumulation (6/16 of the float32 MFMA time; per GEMM at least as close to
 * the float64 product as the float32 MFMA kernels, tests/test_gpu_x6.py).  The 256 x 256 weight-gradient products take their
 * float32 operands through the same split (csrc/wgrad_x6_kernels.hip; PXO_TUNE_X6_WGRAD = 0 keeps them on the float32 pipe);
 * saved tensors, gradients and every other kernel stay float32; pxo_packed_sizes / pxo_pack_weights then describe / write the
 * three-part images (1.5 x the size), both directions.  Opt-in: the headline throughput and every roofline figure of bench.py are PXO_MLP_F32. */
#define PXO_MLP_F32 0
#define PXO_MLP_BF16X3 1
#define PXO_MLP_BF16X6 2

/* One leaf of the parameter arena (offsets in floats, relative to ONE MLP's sub-arena). */
typedef struct PxoLeaf {
  int32_t layer;      /* Dense_<layer> */
  int32_t is_bias;
  int64_t offset;
  int32_t rows;       /* kernel: in ; bias: out */
  int32_t cols;       /* kernel: out; bias: 1   */
} PxoLeaf;

/* ABI version of this header: bumped whenever a struct gains a field or an entry point changes meaning (5: PxoCfg has
 * noise_std + skip_zero_rows, pxo_profile_enable takes a tag MASK, pxo_set_tuning / pxo_occupy_cus exist; 6: PXO_MLP_BF16X6,
 * pxo_adam_pack_step serves every precision, PXO_TUNE_COARSE_REVERSE_STREAM / PXO_TUNE_X6_WGRAD).  A binding checks
 * pxo_version() == PXO_ABI_VERSION and pxo_cfg_bytes() == sizeof(PxoCfg) after dlopen (plenoctree_amd/_lib.py does): a
 * caller built against an older header would otherwise pass a short PxoCfg and have its tail read from past the end. */
#define PXO_ABI_VERSION 6
const char* pxo_last_error(void);
int pxo_version(void);
size_t pxo_cfg_bytes(void);
/* Rows (samples) per tile of the fused MLP kernels in this build (64 or 128); informational. */
int pxo_tile_rows(void);

/* Parameter arena description (flax pytree flattened; replaces the pytree walk of
 * octree/nerf/models.py:75-102).  leaves must hold PXO_NUM_LEAVES entries. */
int pxo_param_layout(const PxoCfg* cfg, PxoLeaf* leaves, int64_t* floats_per_mlp);

/* Sizes (in floats) of the MFMA-fragment-ordered weight images of ONE MLP. */
int pxo_packed_sizes(const PxoCfg* cfg, int64_t* fwd_floats, int64_t* bwd_floats);
/* Re-order one MLP's parameters into the fragment order the fused kernels stream.
 * Must be re-run after every parameter update.  bwd image may be NULL (inference). */
int pxo_pack_weights(const PxoCfg* cfg, const float* mlp_params, float* packed_fwd,
                     float* packed_bwd, void* stream);

/* ---- stage-level entry points (unit-parity hooks against the oracle) -------------- */

/* sample_along_rays + cast_rays, nerf_sh/nerf/model_utils.py:104-142,:97-101.
 * t_rand [B,S] in [0,1) or NULL (randomized=False). z_vals [B,S], pts [B*S,3]. */
int pxo_sample_along_rays(const float* origins, const float* directions, int64_t B, int S,
                          float near_, float far_, int lindisp, const float* t_rand,
                          float* z_vals, float* pts, void* stream);

/* posenc(x, 0, 10), nerf_sh/nerf/model_utils.py:145-173. enc [N,63]. */
int pxo_posenc(const float* x, int64_t N, float* enc, void* stream);

/* Fused posenc + MLP.__call__ (nerf_sh/nerf/model_utils.py:43-94, condition=None;
 * torch twin octree/nerf/model_utils.py:87-158) over M points.
 * raw_rgb [M,3K] may be NULL (sigma only).  Training outputs, all set or all NULL:
 *   acts      8 x [M,256] post-ReLU activations (inputs of the weight-gradient GEMMs)
 *   enc       [M,64] encoded inputs (63 + one zero column)
 *   relu_mask opaque 1-bit/activation image in the kernels' fragment order,
 *             pxo_relu_mask_bytes(M) bytes; consumed by pxo_mlp_bwd_data. */
size_t pxo_relu_mask_bytes(int64_t M);
int pxo_mlp_fwd(const PxoCfg* cfg, const float* packed_fwd, const float* pts, int64_t M,
                float* raw_rgb, float* raw_sigma, float* acts, float* enc, void* relu_mask,
                void* stream);

/* Reverse of pxo_mlp_fwd w.r.t. the trunk (jax.value_and_grad, nerf_sh/train.py:116): writes
 * dz (8 x [M,256], gradient at each layer's pre-activation) and per-tile bias-gradient
 * partials (pxo_dbias_partial_bytes(M) bytes) for pxo_mlp_bwd_weights. */
size_t pxo_dbias_partial_bytes(int64_t M);
int pxo_mlp_bwd_data(const PxoCfg* cfg, const float* packed_bwd, const float* d_raw_rgb,
                     const float* d_raw_sigma, const void* relu_mask, int64_t M, float* dz,
                     float* dbias_partial, void* stream);

/* Parameter gradients of one MLP from saved activations and dz; grads has the layout of
 * one MLP's sub-arena and is overwritten.  ws: pxo_wgrad_workspace_bytes(). */
int pxo_wgrad_workspace_bytes(const PxoCfg* cfg, int64_t M, size_t* bytes);
int pxo_mlp_bwd_weights(const PxoCfg* cfg, const float* acts, const float* enc, const float* dz,
                        const float* d_raw_rgb, const float* d_raw_sigma,
                        const float* dbias_partial, int64_t M, float* grads, void* ws,
                        size_t ws_bytes, void* stream);

/* eval_sh + sigmoid + relu + volumetric_rendering (nerf_sh/nerf/sh.py:54-109,
 * nerf_sh/nerf/models.py:269-284, nerf_sh/nerf/model_utils.py:176-222) for B rays of S
 * samples.  Outputs comp_rgb [B,3], disp [B], acc [B], weights [B,S]. */
int pxo_shade_composite_fwd(const PxoCfg* cfg, const float* raw_rgb, const float* raw_sigma,
                            const float* z_vals, const float* directions, const float* viewdirs,
                            int64_t B, int S, float* comp_rgb, float* disp, float* acc,
                            float* weights, void* stream);
/* Its reverse for a loss that depends on comp_rgb only (nerf_sh/train.py:89-98):
 * d_comp_rgb [B,3] -> d_raw_rgb [B*S,3K], d_raw_sigma [B*S]. */
int pxo_shade_composite_bwd(const PxoCfg* cfg, const float* raw_rgb, const float* raw_sigma,
                            const float* z_vals, const float* directions, const float* viewdirs,
                            const float* d_comp_rgb, int64_t B, int S, float* d_raw_rgb,
                            float* d_raw_sigma, void* stream);

/* The two above and the pixel loss between them (nerf_sh/train.py:89-98: mean((rgb - pixels)^2) over B*3, whose
 * gradient 2 (rgb - pixels) / (3B) seeds the reverse pass) in ONE launch -- what pxo_train_fwd_bwd runs per pass.
 * Outputs: ray_sse [B] = sum_c (rgb_c - pixel_c)^2 per ray, d_raw_rgb [(B*S + n_sp),3K], d_raw_sigma [B*S + n_sp];
 * comp_rgb [B,3] and weights [B,S] may be NULL.  n_sp > 0: rows B*S .. B*S+n_sp-1 of raw_sigma are the sparsity
 * points of train.py:77-85 (loss_sp = w (1 - mean(exp(-len relu(sigma))))): their sigma gradient is written, their
 * rgb gradient zeroed, and sp_exp [n_sp] receives exp(-len relu(sigma)) per point. */
int pxo_shade_composite_train(const PxoCfg* cfg, const float* raw_rgb, const float* raw_sigma,
                              const float* z_vals, const float* directions, const float* viewdirs,
                              const float* pixels, int64_t B, int S, float* comp_rgb, float* weights,
                              float* ray_sse, float* d_raw_rgb, float* d_raw_sigma, int64_t n_sp,
                              float* sp_exp, void* stream);

/* sample_pdf (piecewise_constant_pdf + sort + cast_rays), nerf_sh/nerf/model_utils.py:225-314
 * with bins/weights derived as in nerf_sh/nerf/models.py:296-301.
 * z_coarse, w_coarse [B,Nc]; u [B,Nf] in [0,1) or NULL (randomized=False).
 * z_out [B,Nc+Nf] ascending, pts [B*(Nc+Nf),3]. */
int pxo_sample_pdf(const float* z_coarse, const float* w_coarse, const float* origins,
                   const float* directions, int64_t B, int Nc, int Nf, const float* u,
                   float* z_out, float* pts, void* stream);

/* add_gaussian_noise (nerf_sh/nerf/model_utils.py:317-332; call sites nerf_sh/nerf/models.py:258-264,318-324, between the
 * MLP and the sigma activation): raw[i] += noise_std * n_i in place, i < n.  noise [n]: explicit standard-normal draws
 * (parity runs: the jax key replaced by its draw) or NULL: n_i from Philox stream (seed, stream_id) by Box-Muller
 * (block q = counter, words (0,1) -> elements 4q, 4q+1, words (2,3) -> 4q+2, 4q+3; u1 = ((w >> 8) + 1) / 2^24 in (0,1],
 * u2 = (w' >> 8) / 2^24: z = sqrt(-2 ln u1) (cos, sin)(2 pi u2)).  The whole-path entry points apply it to the raw sigma of
 * the ray samples (streams 3 = coarse, 4 = fine) when PxoCfg.noise_std > 0 and randomized != 0 -- the reference's
 * `(noise_std is not None) and randomized`; the sparsity points (eval_points_raw) get none, as in the reference. */
int pxo_add_gaussian_noise(float* raw, int64_t n, float noise_std, const float* noise, uint64_t seed,
                           uint64_t stream_id, void* stream);

/* Counter-based uniform generator (Philox4x32-10) replacing jax.random.uniform call sites
 * (nerf_sh/nerf/model_utils.py:135,262; nerf_sh/train.py:79).  out[i] in [lo,hi). */
int pxo_uniform(uint64_t seed, uint64_t stream_id, int64_t n, float lo, float hi, float* out,
                void* stream);

/* Uniform integers in [0, n) from the same counter-based generator (np.random.randint in
 * Dataset._next_train, nerf_sh/nerf/datasets.py:159-166). */
int pxo_randint(uint64_t seed, uint64_t stream_id, int64_t count, int64_t n, int64_t* out, void* stream);

/* generate_rays (nerf_sh/nerf/utils.py:545-589, pinhole branch) for B pixels of one camera:
 * c2w = first 3 rows of the 4x4 camera-to-world matrix (12 floats, row-major, device), pixel id
 * p -> (x = p % W, y = p / W); pixel_ids may be NULL (ids 0..B-1 = a whole image). */
int pxo_generate_rays(const float* c2w, int W, int H, float focal, const int64_t* pixel_ids, int64_t B,
                      float* origins, float* directions, float* viewdirs, void* stream);

/* Dataset._next_train for one image (nerf_sh/nerf/datasets.py:159-166: `ray_indices = np.random.randint(0, H*W, (B,))`, the
 * rays and the pixels of those indices) in one launch: pixel id i = pxo_randint's element i of stream (seed, stream_id)
 * mod W*H, its ray as pxo_generate_rays, its colour from image_rgb [H*W,3] (the resident training image) -- bit for bit what
 * the three separate calls give.  pixel_ids [B] may be NULL.  `first`: the B elements are elements first .. first + B - 1
 * of the stream -- a rank's shard of ONE global draw (the reference's single-host step: one image, its batch_size pixels
 * sharded over the local devices, nerf_sh/nerf/datasets.py:159-166 + nerf_sh/nerf/utils.py:518-522); 0 = the whole draw. */
int pxo_sample_batch(uint64_t seed, uint64_t stream_id, const float* c2w, int W, int H, float focal,
                     const float* image_rgb, int64_t B, int64_t first, int64_t* pixel_ids, float* origins,
                     float* directions, float* viewdirs, float* pixels, void* stream);

/* The same for the `image_batching` sampler (nerf_sh/nerf/datasets.py:137-141,152-157: rays of ALL
 * images flattened into one table): c2w [n_cams,3,4], ray id r -> camera r / (W*H), pixel r % (W*H). */
int pxo_generate_rays_multi(const float* c2w, int n_cams, int W, int H, float focal, const int64_t* ray_ids,
                            int64_t B, float* origins, float* directions, float* viewdirs, void* stream);

/* Step 2 of the extraction (octree/extraction.py:391-393, SH/SG formats): mean over the S samples
 * of each leaf of cat([raw_rgb, raw_sigma]); out [n_cells, 3K+1]. */
int pxo_mean_over_samples(const PxoCfg* cfg, const float* raw_rgb, const float* raw_sigma, int64_t n_cells,
                          int S, float* out, void* stream);

/* flax.optim.Adam.apply_gradient (call site nerf_sh/train.py:119; beta1 .9, beta2 .999,
 * eps 1e-8), with g = grads*grad_scale (grad_scale = 1/world_size after an RCCL sum).
 * `step` = number of updates already applied. */
int pxo_adam_step(float* params, float* m, float* v, const float* grads, int64_t n, float lr,
                  int64_t step, float grad_scale, void* stream);

/* The same update of the whole 2-MLP arena AND the refresh of the four fragment-ordered images in one launch
 * (state.optimizer.apply_gradient + the re-pack that must follow it): equals pxo_adam_step followed by
 * pxo_pack_weights on both MLPs, bit for bit.  The images must have been written by pxo_pack_weights once (their
 * zero padding is not rewritten); packed_bwd0/1 may both be NULL.  With a split-precision cfg the same call runs the Adam
 * kernel and then the packing kernels of that precision (two or three launches instead of one, same results as the separate
 * calls). */
int pxo_adam_pack_step(const PxoCfg* cfg, float* params, float* m, float* v, const float* grads, float lr,
                       int64_t step, float grad_scale, float* packed_fwd0, float* packed_bwd0,
                       float* packed_fwd1, float* packed_bwd1, void* stream);

/* ---- whole-path entry points ------------------------------------------------------ */

/* NerfModel.__call__ forward only (nerf_sh/nerf/models.py:216-348; called by
 * get_render_pfn nerf_sh/nerf/utils.py:701-713).  packed_fwd0/1: images of MLP_0/MLP_1.
 * t_rand/u as above (NULL with randomized!=0 draws them from `seed`).
 * Outputs [B,3],[B],[B] for coarse and fine (fine ones may be NULL if num_fine_samples==0). */
int pxo_render_workspace_bytes(const PxoCfg* cfg, int64_t B, size_t* bytes);
int pxo_render_fwd(const PxoCfg* cfg, const float* packed_fwd0, const float* packed_fwd1,
                   const float* origins, const float* directions, const float* viewdirs,
                   int64_t B, int randomized, const float* t_rand, const float* u, uint64_t seed,
                   float* rgb_c, float* disp_c, float* acc_c, float* rgb_f, float* disp_f,
                   float* acc_f, void* ws, size_t ws_bytes, void* stream);

/* loss_fn + value_and_grad of train_step (nerf_sh/train.py:66-116) on this device's shard.
 * params: the 2-MLP arena; packed_*: its images (pxo_pack_weights).  grads: 2-MLP arena,
 * overwritten with d(total loss)/d(params).  stats[6] (device) = loss, psnr, loss_c,
 * loss_sp, psnr_c, weight_l2 (Stats, nerf_sh/nerf/utils.py:43-50).
 * sp_points [npoints,3] or NULL (drawn from seed).  The cross-device mean (pmean,
 * train.py:117-118) and Adam are the caller's next two steps. */
int pxo_train_workspace_bytes(const PxoCfg* cfg, int64_t B, size_t* bytes);
int pxo_train_fwd_bwd(const PxoCfg* cfg, const float* params, const float* packed_fwd0,
                      const float* packed_bwd0, const float* packed_fwd1,
                      const float* packed_bwd1, const float* origins, const float* directions,
                      const float* viewdirs, const float* pixels, int64_t B, int randomized,
                      const float* t_rand, const float* u, const float* sp_points, uint64_t seed,
                      float* grads, float* stats, void* ws, size_t ws_bytes, void* stream);

/* The same, for data-parallel training with the gradient exchange split in two buckets.  The kernels are sequenced
 * coarse level first -- forward, losses, reverse through MLP_0 -- and only then the fine level (the reference's single
 * jax.value_and_grad, nerf_sh/train.py:116, leaves the order to XLA; no gradient of the fine level reaches MLP_0 because
 * the fine sample positions are stop_gradient'ed, nerf_sh/nerf/model_utils.py:286).  `grads0_ready` (a hipEvent_t, e.g.
 * from pxo_event_create; may be NULL) is recorded on `stream` at the point where grads[0 .. n_mlp) -- MLP_0's
 * sub-arena, weight decay included -- is final: a second stream that waits for it can run lax.pmean of that half
 * (train.py:117) under the fine level, while grads[n_mlp .. 2 n_mlp) and stats follow at the end of the call.
 * pxo_train_fwd_bwd is this function with grads0_ready = NULL. */
int pxo_train_fwd_bwd_bucketed(const PxoCfg* cfg, const float* params, const float* packed_fwd0,
                               const float* packed_bwd0, const float* packed_fwd1,
                               const float* packed_bwd1, const float* origins, const float* directions,
                               const float* viewdirs, const float* pixels, int64_t B, int randomized,
                               const float* t_rand, const float* u, const float* sp_points, uint64_t seed,
                               float* grads, float* stats, void* ws, size_t ws_bytes, void* grads0_ready,
                               void* stream);

/* How much of the last pxo_train_fwd_bwd call's reverse pass was live: 16-row chunks with a non-zero upstream gradient
 * out of all chunks of both levels (with cfg->skip_zero_rows = 0 both numbers are the total).  Reads the flags the call
 * left in `ws`; synchronises `stream`.  Reporting only (bench.py's `converge` record, tests). */
int pxo_train_backward_work(const PxoCfg* cfg, int64_t B, void* ws, size_t ws_bytes, int64_t* live_chunks,
                            int64_t* total_chunks, void* stream);

/* Plain event handles for the call above (hipEventDisableTiming), so that a host runtime whose own event objects are
 * lazily created or private (torch.cuda.Event) can still order its collective stream behind the bucket:
 * pxo_stream_wait_event(side_stream, ev) = "side_stream continues once the last record of ev has completed". */
int pxo_event_create(void** event);
int pxo_event_destroy(void* event);
int pxo_stream_wait_event(void* stream, void* event);

/* NerfModel.eval_points_raw (octree/nerf/models.py:211-252; call sites
 * octree/extraction.py:271,316,373).  raw_rgb may be NULL (step1 keeps sigma only). */
int pxo_eval_points(const PxoCfg* cfg, const float* packed_fwd, const float* points, int64_t N,
                    float* raw_rgb, float* raw_sigma, void* stream);

/* Dense-grid driver of octree/extraction.py:290-320 (step1) / :250-274 (auto_scale):
 * evaluates sigma at grid point ((i+.5)/reso - offset[a]) / scale[a] for x in [x0,x1),
 * all y,z (ij meshgrid order, x slowest).  sigma_out [(x1-x0)*reso*reso]. */
int pxo_grid_sigma(const PxoCfg* cfg, const float* packed_fwd, int reso, int x0, int x1,
                   const float offset[3], const float scale[3], float* sigma_out, void* stream);

/* ---- run-time choices between implementations of the same result ----------------------- */
/* Process-wide, and not synchronised with steps in flight on other host threads: a step reads each knob once when it is
 * enqueued (and records what it decided: pxo_train_backward_work reports for the step that ran, not for today's knobs).
 * The tile schedule leaves every bit unchanged; the split-K ranges change the (still fixed) order of the
 * weight-gradient sums, i.e. float32 round-off (tests/test_gpu_parity.py holds both).  Used by A/B sessions (bench.py --tune)
 * and equality tests; nothing is read from the environment.
 *   PXO_TUNE_TILE_SCHED    how the persistent workgroups of the dense training kernels (mlp_fwd with saved tensors,
 *                          mlp_bwd_data) pick their 128-row tiles inside pxo_train_fwd_bwd*: 0 = static stride,
 *                          1 (default) = from a device counter, so that a workgroup that starts late -- because a
 *                          collective's kernel held its CU at the launch boundary -- is not the launch's tail.  (The
 *                          zero-row skipping backward always uses the counter.)
 *   PXO_TUNE_WGRAD_RANGES  row ranges (split-K slabs) per layer of the 256x256 weight-gradient products: 0 = built-in
 *                          choice by pass size, n = exactly n (1 .. number of CUs). */
#define PXO_TUNE_TILE_SCHED 0
#define PXO_TUNE_WGRAD_RANGES 1
#define PXO_TUNE_WGRAD_SKINNY_RANGES 2   /* the same for the two skinny products (enc-based pair, heads): 1 .. 2 x number of CUs */
/*   PXO_TUNE_COARSE_REVERSE_STREAM  0 (default): every launch of pxo_train_fwd_bwd* on the caller's stream.  1: the reverse pass of
 *                          the coarse level runs on an internal low-priority side stream (created on first use) beside the fine
 *                          level's forward and joins the caller's stream before the fine weight gradients; `grads0_ready` is
 *                          then recorded on that side stream.  Bits unchanged; +0.2 % / +0.7 % at 512 / 4096 rays (r06b). */
#define PXO_TUNE_COARSE_REVERSE_STREAM 3
/*   PXO_TUNE_X6_WGRAD      with PXO_MLP_BF16X6 only.  1 (default): the 256x256 weight-gradient products of Dense_1..7 run in bf16x6
 *                          too (wgrad_x6_kernels.hip); 0: on the float32 MFMA pipe like the float32 path (A/B). */
#define PXO_TUNE_X6_WGRAD 4
int pxo_set_tuning(int knob, int value);
int pxo_get_tuning(int knob, int* value);

/* ---- measurement ------------------------------------------------------------------ */
/* Diagnostic for contention probes (scripts/contention_probe.py): `blocks` workgroups of `threads` threads that idle for
 * `micros` microseconds of the device's constant-rate clock on `stream` -- what the kernel of a ring all-reduce looks like to
 * the kernels it shares the GPU with (it occupies CU slots and moves no data).  lds_bytes (0 .. 65536) of dynamic LDS per
 * workgroup: 0 = a workgroup that fits beside a resident fused-MLP workgroup (133 KB of the CU's 160 KB LDS, 368 of 512
 * registers per SIMD lane), 64 KB = one that does not and must wait for a CU to become free.  No reference counterpart. */
int pxo_occupy_cus(int blocks, int threads, float micros, int lds_bytes, void* stream);

/* HIP-event timing of the dominant kernels on the stream they are launched on (bench.py's
 * roofline leg; the reference only has wall-clock rays/sec, nerf_sh/train.py:222-226).
 * Tags: 0 mlp_fwd, 1 mlp_bwd_data, 2 wgrad 256x256 GEMM, 3 other wgrad GEMMs.
 * pxo_profile_enable(mask): bit t set = launches tagged t are bracketed by two event records; 0 = off,
 * PXO_PROF_ALL = every tag.  An event record costs the stream ~5 us (a barrier packet between two kernels that
 * would otherwise dispatch back to back: measured, profiles/r04*), so a timed region that must stay representative
 * enables only the tag it needs. */
#define PXO_PROF_MLP_FWD 0
#define PXO_PROF_MLP_BWD_DATA 1
#define PXO_PROF_WGRAD_MAIN 2
#define PXO_PROF_WGRAD_OTHER 3
#define PXO_PROF_NUM_TAGS 4
#define PXO_PROF_ALL ((1 << PXO_PROF_NUM_TAGS) - 1)
int pxo_profile_enable(int tag_mask);
/* Synchronises the recorded events and returns launches / total ms / total rows processed
 * for `tag` since the last read; resets the tag. */
int pxo_profile_read(int tag, int64_t* launches, double* total_ms, int64_t* total_rows);

#ifdef __cplusplus
}
#endif
#endif /* PLENOCTREE_HIP_H_ */
