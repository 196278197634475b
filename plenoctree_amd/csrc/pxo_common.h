// Shared host/device definitions for the gfx950 NeRF-SH kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/plenoctree_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace pxo {

constexpr int kW = PXO_NET_WIDTH;        // 256
constexpr int kDepth = PXO_NET_DEPTH;    // 8
constexpr int kEnc = PXO_ENC_DIM;        // 63
constexpr int kEncPad = PXO_ENC_PAD;     // 64
// Geometry of the fused MLP kernels: 128-row tiles, 8 waves (every wave owns all rows x 32 columns of a layer),
// one persistent workgroup per CU.  One weight-fragment load then feeds 16 MFMAs (scripts/ubench/gemm_geom.hip:
// 149.7 TFLOP/s for the bare loop against 138.3 for 64 x 64 per wave).
constexpr int kTM = 128;
constexpr int kMlpThreads = 512;
constexpr int kMlpWgPerCu = 1;
constexpr int kLDA = 260;                // LDS row stride (floats): 256 + one b128 access of pad
constexpr int kMlpWaves = kMlpThreads / 64;
constexpr int kMaskWords = (kTM / 32) * (8 / kMlpWaves) * 16 / 32;  // relu-mask words per thread per layer
constexpr int kMaxMlpGrid = 1024;        // upper bound on persistent workgroups
constexpr int kLiveRows = 16;            // rows per "live" flag of the zero-row skipping backward (= the wgrad chunk height)
constexpr int kMaxLiveChunks = 2048;     // live-chunk list of a SPARSE wgrad workgroup (LDS): row ranges of at most 32,768 rows

// ---- derived sizes -----------------------------------------------------------------
__host__ __device__ inline int sh_dim(int deg) { return (deg + 1) * (deg + 1); }
__host__ __device__ inline int rgb_channels(int deg) { return 3 * sh_dim(deg); }
// head width padded to 32-col MFMA blocks: cols [0,3K) = Dense_9 (rgb), col 3K = Dense_8 (sigma)
__host__ __device__ inline int head_blocks(int deg) { return (rgb_channels(deg) + 1 + 31) / 32; }

// input rows of Dense_l (l = 0..9) in the reference layout
__host__ __device__ inline int layer_in(int l) {
  return l == 0 ? kEnc : (l == 5 ? kW + kEnc : kW);
}
__host__ __device__ inline int layer_out(int l, int deg) {
  return l < 8 ? kW : (l == 8 ? 1 : rgb_channels(deg));
}
// offset (floats) of Dense_l kernel inside ONE MLP's sub-arena; bias follows its kernel
__host__ __device__ inline int64_t leaf_kernel_off(int l, int deg) {
  int64_t off = 0;
  for (int i = 0; i < l; ++i) off += (int64_t)layer_in(i) * layer_out(i, deg) + layer_out(i, deg);
  return off;
}
__host__ __device__ inline int64_t leaf_bias_off(int l, int deg) {
  return leaf_kernel_off(l, deg) + (int64_t)layer_in(l) * layer_out(l, deg);
}
__host__ __device__ inline int64_t mlp_param_count(int deg) { return leaf_kernel_off(10, deg); }

// ---- packed (MFMA fragment order) weight images --------------------------------------
// A packed matrix B[K][N] (K = 8*KG, N = 32*NCB) stores element (k,n) at
//   ((g*NCB + c)*64 + lane)*4 + j,  g=k/8, lane=((k%8)/4)*32 + n%32, j=k%4, c=n/32
// so that one wave reads a (k-group, col-block) fragment as 64 lanes x 16 B contiguous and
// lane l feeds mfma_f32_32x32x2f32 number j with B[k=8g+4*(l>>5)+j][n=32c+(l&31)].
__host__ __device__ inline int64_t packed_index(int k, int n, int ncb) {
  int g = k >> 3, kk = k & 7;
  int lane = ((kk >> 2) << 5) | (n & 31);
  return (((int64_t)g * ncb + (n >> 5)) * 64 + lane) * 4 + (kk & 3);
}

// forward image: trunk layer l (0..7) then heads then biases
__host__ __device__ inline int fwd_kgroups(int l) { return l == 0 ? 8 : (l == 5 ? 40 : 32); }
__host__ __device__ inline int64_t fwd_layer_off(int l) {  // l in 0..8 (8 = heads)
  int64_t off = 0;
  for (int i = 0; i < l; ++i) off += (int64_t)fwd_kgroups(i) * 8 * 256;
  return off;
}
__host__ __device__ inline int64_t fwd_bias_off(int deg) {
  return fwd_layer_off(8) + (int64_t)32 * head_blocks(deg) * 256;
}
__host__ __device__ inline int64_t fwd_image_floats(int deg) {
  return fwd_bias_off(deg) + 8 * kW + 32 * head_blocks(deg);
}
// backward(data) image: head^T (K = 32*NHB, N = 256), then W_l^T for l = 7..1 (K = 256 outputs,
// N = first 256 inputs)
__host__ __device__ inline int64_t bwd_layer_off(int l, int deg) {  // l in 1..7
  return (int64_t)head_blocks(deg) * 32 * 256 + (int64_t)(7 - l) * 256 * 256;
}
__host__ __device__ inline int64_t bwd_image_floats(int deg) { return bwd_layer_off(0, deg); }

// ---- bf16x6 images (mlp_x6_kernels.hip; PxoCfg.mlp_precision = PXO_MLP_BF16X6) ---------------------------------------
// Every weight is split into three bf16 parts w = w1 + w2 + w3; a (16-deep k-group, 32-column block) is stored as three
// fragments [w1 | w2 | w3] of 64 lanes x 16 B (8 bf16: k = 16 kg + 8 (lane >> 5) + e, column 32 cb + (lane & 31)).
// Offsets below are in 4-byte slots, like the float32 images'.
constexpr int kX6KgSlots = 8 * 3 * 256;       // slots per k-group of a 256-column layer
__host__ __device__ inline int x6_fwd_kg(int l) { return l == 0 ? 4 : (l == 5 ? 20 : 16); }
__host__ __device__ inline int64_t x6_fwd_layer_off(int l) {  // l in 0..8 (8 = heads)
  int64_t off = 0;
  for (int i = 0; i < l; ++i) off += (int64_t)x6_fwd_kg(i) * kX6KgSlots;
  return off;
}
__host__ __device__ inline int64_t x6_fwd_bias_off(int deg) {
  return x6_fwd_layer_off(8) + (int64_t)16 * head_blocks(deg) * 3 * 256;
}
__host__ __device__ inline int64_t x6_fwd_image_floats(int deg) {
  return x6_fwd_bias_off(deg) + 8 * kW + 32 * head_blocks(deg);
}
// backward(data): head^T (K = the head columns, zero-padded to whole groups of 4 k-groups), then W_l^T for l = 7..1
__host__ __device__ inline int x6_bwd_head_kg(int deg) { return 4 * ((head_blocks(deg) + 1) / 2); }
__host__ __device__ inline int64_t x6_bwd_image_floats(int deg) {
  return (int64_t)(x6_bwd_head_kg(deg) + 7 * 16) * kX6KgSlots;
}

__host__ __device__ inline int64_t num_tiles(int64_t M) { return (M + kTM - 1) / kTM; }

// Tile schedule of one fused-MLP launch over M rows on `grid` persistent workgroups (the same function of (M, grid)
// in the forward and the backward(data) kernel): whole rounds of full 128-row tiles; if the rows that are left fit
// into ONE round at half height they are cut into 64-row tiles (one per workgroup, 2 of the 4 row blocks), so the
// ragged last round costs about half a tile time instead of a whole one (4096 rays x 192 samples + 10,000 sparsity
// points = 24 full rounds + 157 half tiles on 256 CUs, instead of 25 rounds).  Tile t lives in mask/partial slot t;
// half tile h in slot n_full + h.
struct TileSched {
  int64_t n_full;      // full tiles: rows [t*128, t*128+128)
  int64_t n_half;      // half tiles: rows [half_row0 + h*64, +64)
  int64_t half_row0;
};
__host__ __device__ inline TileSched tile_sched(int64_t M, int64_t grid) {
  TileSched t;
  const int64_t tiles = num_tiles(M);
  t.n_full = tiles; t.n_half = 0; t.half_row0 = tiles * kTM;
  if (grid < 1 || tiles <= grid) return t;
  const int64_t whole = (tiles / grid) * grid;
  const int64_t left = M - whole * kTM;             // > 0 rows after the whole rounds (0 if tiles % grid == 0 and M % 128 == 0)
  if (left > 0 && left <= grid * (kTM / 2)) {
    t.n_full = whole;
    t.half_row0 = whole * kTM;
    t.n_half = (left + kTM / 2 - 1) / (kTM / 2);
  }
  return t;
}
// relu-mask image written by the forward kernel: per (slot, layer, thread) kMaskWords words
__host__ __device__ inline int64_t mask_slots(int64_t M) { return num_tiles(M) + kMaxMlpGrid / 2; }
__host__ __device__ inline int64_t mask_words(int64_t M) {
  return mask_slots(M) * kDepth * kMlpThreads * kMaskWords;
}
// bias-gradient partials written by the backward-data kernel, one per TILE SLOT: [slot][9][256] (a tile's column sums do
// not depend on which workgroup computed it, so any tile schedule gives the same bits), + one live byte per slot behind
__host__ __device__ inline int64_t dbias_floats(int64_t M) { return mask_slots(M) * (9 * kW + 1); }
__host__ __device__ inline const uint8_t* dbias_tile_live(const float* dbias_partial, int64_t M) {
  return reinterpret_cast<const uint8_t*>(dbias_partial + mask_slots(M) * 9 * kW);
}

// ---- error plumbing ------------------------------------------------------------------
void set_error(const char* fmt, ...);
int check_launch(const char* what);

#define PXO_REQUIRE(cond, ...)            \
  do {                                    \
    if (!(cond)) {                        \
      pxo::set_error(__VA_ARGS__);        \
      return PXO_ERR_ARG;                 \
    }                                     \
  } while (0)

int validate_cfg(const PxoCfg* cfg);
int num_cus();
int mlp_bwd_partials(int64_t M);   // number of [9][256] tile partials (slots) mlp_bwd_data writes for M rows

// HIP-event bracket around one kernel launch (active for the tags in pxo_profile_enable's mask)
struct KernelTimer {
  KernelTimer(int tag, int64_t rows, hipStream_t s);
  ~KernelTimer();
  int slot;
  hipStream_t stream;
};
int launch_mlp_fwd_grid(const PxoCfg* cfg, const float* packed_fwd, int reso, int x0, int x1,
                        const float* off, const float* scale, float* sigma_out, hipStream_t s);

// ---- launchers implemented in the kernel translation units -----------------------------
int launch_pack(const PxoCfg* cfg, const float* mlp_params, float* fwd, float* bwd, hipStream_t s);
// tile_counter (may be NULL = static tile stride): a ZERO device word; the persistent workgroups of the training
// instantiation then take their tiles from it (mlp_kernels.hip TileTicket)
int launch_mlp_fwd(const PxoCfg* cfg, const float* packed_fwd, const float* pts, int64_t M,
                   float* raw_rgb, float* raw_sigma, float* acts, float* enc, uint32_t* mask,
                   hipStream_t s, unsigned int* tile_counter = nullptr);
// chunk_live (may be NULL = dense): one byte per kLiveRows rows, WRITTEN by the backward(data) kernel (1: some row of the
// chunk has a non-zero upstream gradient) and READ by the weight-gradient kernels, which skip dead chunks
__host__ __device__ inline int64_t live_flags(int64_t M) { return (M + kLiveRows - 1) / kLiveRows + kTM / kLiveRows; }
// tile_counter (may be NULL = static tile stride): a device word; the persistent workgroups then TAKE tiles from it
// instead of striding over them (skipped tiles do not leave workgroups idle; a late-starting workgroup is not the tail).
// counter_is_zero: the caller zeroed it on the stream already (the train step's first launch does); otherwise a memset is
// enqueued here.
int launch_mlp_bwd_data(const PxoCfg* cfg, const float* packed_bwd, const float* d_raw_rgb,
                        const float* d_raw_sigma, const uint32_t* mask, int64_t M, float* dz,
                        float* dbias_partial, uint8_t* chunk_live, unsigned int* tile_counter, hipStream_t s,
                        bool counter_is_zero = false, int flags = 0);
// flags of launch_mlp_bwd_data / launch_mlp_bwd_weights.  kBiasFromWgrad (bf16x6 with its own weight-gradient kernel only; the
// train step sets it on BOTH launches of a pass): the bias gradients of Dense_1..7 are the column sums of dz_1..7 that
// wgrad_x6_kernel takes while it streams dz -- backward(data) then skips its per-layer lane reductions for those layers
// (6 % of its time) and the reduce reads the per-range sums instead of the per-slot partials.
constexpr int kBiasFromWgrad = 1;
// run-time choices between implementations of the same result (pxo_set_tuning; A/B sessions and equality tests)
int tune_tile_sched();        // PXO_TUNE_TILE_SCHED: 0 static stride, 1 device counter (dense training kernels)
int tune_wgrad_ranges();      // PXO_TUNE_WGRAD_RANGES: 0 = built-in choice, n > 0 = row ranges per layer of the 256x256 products
int tune_wgrad_skinny_ranges();  // PXO_TUNE_WGRAD_SKINNY_RANGES: the same for the enc-based pair and the head product
int tune_x6_wgrad();          // PXO_TUNE_X6_WGRAD: 1 (default) = in bf16x6 the 256x256 weight gradients run on the bf16 pipe too, 0 = float32 MFMA
// can the weight-gradient kernels skip dead chunks for a pass of M rows (every row range fits a workgroup's live list)?
// If not the whole reverse pass of the step runs dense (pxo_train_fwd_bwd decides up front).
bool wgrad_skip_supported(int64_t M);
// wgrad_x6_kernels.hip: the 256x256 products of Dense_1..7 in bf16x6 (same grid, slabs and reduce as the float32 launch)
void launch_wgrad_main_x6(const float* acts, const float* dz1, int64_t M, int64_t rpw, int P, float* slab, int n_layers,
                          int64_t layer_stride, const uint8_t* chunk_live, float* dz_colsum, hipStream_t s);
size_t wgrad_workspace_bytes(const PxoCfg* cfg, int64_t M);
int launch_mlp_bwd_weights(const PxoCfg* cfg, const float* acts, const float* enc, const float* dz,
                           const float* d_raw_rgb, const float* d_raw_sigma,
                           const float* dbias_partial, int64_t M, float* grads, void* ws,
                           size_t ws_bytes, const uint8_t* chunk_live, hipStream_t s, int flags = 0);
int launch_count_live(const uint8_t* chunk_live, int64_t n, unsigned long long* out, hipStream_t s);
int launch_posenc(const float* x, int64_t N, float* enc, hipStream_t s);
// opt-in split-precision forward (mlp_x3_kernels.hip); pts == nullptr selects the dense-grid point source
int launch_pack_x3(const PxoCfg* cfg, const float* mlp_params, float* fwd, hipStream_t s);
int launch_mlp_fwd_x3(const PxoCfg* cfg, const float* packed_fwd, const float* pts, int reso, int x0, const float* off,
                      const float* scale, int64_t M, float* raw_rgb, float* raw_sigma, hipStream_t s);

// opt-in float32-accurate split-precision training kernels (mlp_x6_kernels.hip); pts == nullptr selects the dense-grid source
int launch_pack_x6(const PxoCfg* cfg, const float* mlp_params, float* fwd, float* bwd, hipStream_t s);
int launch_mlp_fwd_x6(const PxoCfg* cfg, const float* packed_fwd, const float* pts, int reso, int x0, const float* off,
                      const float* scale, int64_t M, float* raw_rgb, float* raw_sigma, float* acts, float* enc,
                      uint32_t* mask, unsigned int* tile_counter, hipStream_t s);
int launch_mlp_bwd_data_x6(const PxoCfg* cfg, const float* packed_bwd, const float* d_raw_rgb, const float* d_raw_sigma,
                           const uint32_t* mask, int64_t M, float* dz, float* dbias_partial, uint8_t* chunk_live,
                           unsigned int* tile_counter, hipStream_t s, bool db_all = true);

int launch_sample_along_rays(const float* o, const float* d, int64_t B, int S, float near_,
                             float far_, int lindisp, const float* t_rand, float* z, float* pts,
                             hipStream_t s);
int launch_shade_composite_fwd(const PxoCfg* cfg, const float* raw_rgb, const float* raw_sigma,
                               const float* z, const float* dirs, const float* viewdirs, int64_t B,
                               int S, float* comp_rgb, float* disp, float* acc, float* weights,
                               hipStream_t s);
int launch_shade_composite_bwd(const PxoCfg* cfg, const float* raw_rgb, const float* raw_sigma,
                               const float* z, const float* dirs, const float* viewdirs,
                               const float* d_comp_rgb, int64_t B, int S, float* d_raw_rgb,
                               float* d_raw_sigma, hipStream_t s);
int launch_sample_pdf(const float* z_c, const float* w_c, const float* o, const float* d, int64_t B,
                      int Nc, int Nf, const float* u, float* z_out, float* pts, hipStream_t s);
int launch_generate_rays(const float* c2w, int n_cams, int W, int H, float focal, const int64_t* pix, int64_t B,
                         float* o, float* d, float* v, hipStream_t s);
int launch_sample_batch(uint64_t seed, uint64_t stream_id, const float* c2w, int W, int H, float focal, const float* image,
                        int64_t B, int64_t first, int64_t* ids, float* o, float* d, float* v, float* pixels, hipStream_t s);
int launch_randint(uint64_t seed, uint64_t stream_id, int64_t count, int64_t n, int64_t* out, hipStream_t s);
int launch_mean_samples(const float* raw_rgb, const float* raw_sigma, int64_t n_cells, int S, int C, float* out,
                        hipStream_t s);
int launch_uniform(uint64_t seed, uint64_t stream_id, int64_t n, float lo, float hi, float* out,
                   hipStream_t s);
int launch_add_noise(float* raw, int64_t n, float noise_std, const float* noise, uint64_t seed, uint64_t stream_id, hipStream_t s);
int launch_adam(float* p, float* m, float* v, const float* g, int64_t n, float lr, int64_t step,
                float grad_scale, hipStream_t s);
// training form of the compositing: forward + pixel loss + reverse in one launch (see render_kernels.hip); serves the
// n_sp sparsity rows appended to the pass as well
int launch_shade_composite_train(const PxoCfg* cfg, const float* raw_rgb, const float* raw_sigma, const float* z,
                                 const float* dirs, const float* viewdirs, const float* pixels, int64_t B, int S,
                                 float* comp_rgb, float* weights, float* ray_sse, float* d_raw_rgb, float* d_raw_sigma,
                                 int64_t n_sp, float* sp_exp, hipStream_t s);
// up to 3 uniform draws (Philox streams of one seed) in one launch
struct UniformJob { uint64_t stream_id; int64_t n; float lo, hi; float* out; };
// sq_x != NULL: the same launch also writes the kSumsqBlocks fixed-order partial sums of squares of sq_x[0 .. sq_n) (the
// parameter norm of weight_l2, train.py:101-108: it depends on the parameters only, so it rides with the step's first launch)
constexpr int kSumsqBlocks = 64;
// zero_words != NULL: the same launch also zeroes n_zero (<= 256) device words (the tile counters of the step's MLP launches)
int launch_uniform_jobs(uint64_t seed, const UniformJob* jobs, int n_jobs, hipStream_t s, const float* sq_x = nullptr,
                        int64_t sq_n = 0, float* sq_partial = nullptr, unsigned int* zero_words = nullptr, int n_zero = 0);
int launch_finalize_stats(const float* sse_f, const float* sse_c, const float* sp_exp, const float* sumsq_partial,
                          int64_t B, int64_t n_sp, float sp_weight, int64_t n_params, float* stats, hipStream_t s);
int launch_adam_pack(const PxoCfg* cfg, float* p, float* m, float* v, const float* g, float lr, int64_t step,
                     float grad_scale, float* fwd0, float* bwd0, float* fwd1, float* bwd1, hipStream_t s);
int launch_fill(float* p, int64_t n, float v, hipStream_t s);
int launch_axpy(float* y, const float* x, int64_t n, float a, hipStream_t s);

}  // namespace pxo
