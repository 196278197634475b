// Opt-in float32-accurate split-precision TRAINING kernels of the NeRF-SH MLP for gfx950: PxoCfg.mlp_precision = PXO_MLP_BF16X6.
//
// The fused forward (with saved tensors) and backward(data) of mlp_kernels.hip with every GEMM evaluated on the bf16 matrix
// pipe: each float32 operand is split EXACTLY into three bf16 parts, x = x1 + x2 + x3 (x1 = bf16(x), x2 = bf16(x - x1),
// x3 = bf16(x - x1 - x2); round to nearest even gives 8 + 8 + 8 significand bits and signed residuals, so the sum is x itself
// for every normal float32), and a product is the six partial products of order <= 2^-16,
//     w1 x1 + (w1 x2 + w2 x1 + w2 x2 + w1 x3 + w3 x1)            dropped: w2 x3, w3 x2, w3 x3 = O(2^-24 |w x|),
// six v_mfma_f32_32x32x16_bf16 with float32 accumulation (bf16 x bf16 products are exact in float32) = 6/16 of the float32
// MFMA time.  The leading product and the five corrections go to SEPARATE accumulators that are added once per layer: the
// 16 roundings of the leading chain (K = 256 / 16) are the only ones at the full magnitude, against 128 in the float32-MFMA
// kernel's chain -- the result is CLOSER to the float64 product than the float32 kernel's (tests/test_gpu_x6.py).
//
// What stays float32: everything that leaves the kernels.  acts / enc / dz / raw_* / bias partials are float32 row-major
// arrays in the layouts of mlp_kernels.hip, so the weight-gradient kernels (the 256x256 products: wgrad_x6_kernels.hip, which
// splits those float32 arrays itself; the skinny ones: wgrad_kernels.hip, native float32 MFMA), the compositing kernels and the
// workspace are untouched; only the relu-mask words are in this file's own order.
//
// Replaces posenc + MLP.__call__ (nerf_sh/nerf/model_utils.py:43-94,145-173) and its reverse (jax.value_and_grad,
// nerf_sh/train.py:116) for the same configurations as mlp_fwd_kernel / mlp_bwd_data_kernel.
//
// Geometry: three bf16 planes of an activation tile are 6 B per element, so 64 rows x 256 (x 264 / 256 padding) = 101 KB is what
// fits the CU's 160 KB of LDS: one persistent 8-wave workgroup per CU walks the SAME 128-row slots as the float32 kernels
// (pxo_common.h TileSched: mask words, bias partials and live flags keep their slot indices) as two 64-row sub-tiles.  The
// product is computed transposed (weights = A operand, activations = B operand): a lane owns 4 consecutive features of one
// sample per register quad, so the epilogue writes one ds_write_b64 per plane and one 16-byte global store per quad.
// Every wave owns 64 rows x 32 features: per 16-deep k-group 6 ds_read_b128 + 3 buffer_load_b128 feed 12 MFMAs.
#include "pxo_common.h"
#include "pxo_x6.h"

namespace pxo {

constexpr int kYRows = 64;                   // rows of a sub-tile
constexpr int kYRB = kYRows / 32;            // row blocks (all owned by every wave)
constexpr int kYThreads = kMlpThreads;       // 512
constexpr int kYWaves = kYThreads / 64;      // 8: wave w owns features [32 w, 32 w + 32)
constexpr int kLDB = 264;                    // LDS row stride in bf16 (256 + 8: conflict-free ds_read_b128)
constexpr int kPlane = kYRows * kLDB;        // bf16 elements per plane
static_assert(kYWaves * 32 == kW, "one 32-feature block per wave");

// ------------------------------------------------------------------------------------------
// weight packing
// ------------------------------------------------------------------------------------------
// W_l[k_in][n_out] of the reference layout with zero padding; l == 8: the fused head (cols [0,C) = Dense_9, col C = Dense_8)
__device__ __forceinline__ float x6_src_weight(const float* __restrict__ p, int deg, int l, int k, int n) {
  const int C = rgb_channels(deg);
  if (l < 8) {
    if (k >= layer_in(l) || n >= kW) return 0.f;
    return p[leaf_kernel_off(l, deg) + (int64_t)k * kW + n];
  }
  if (k >= kW) return 0.f;
  if (n < C) return p[leaf_kernel_off(9, deg) + (int64_t)k * C + n];
  if (n == C) return p[leaf_kernel_off(8, deg) + k];
  return 0.f;
}

// one thread per 4-byte slot = elements (2s, 2s + 1) of a lane's 16-byte fragment of part `part`
// forward: A[feature n][k] = W_l[k][n]
__global__ void pack_fwd_x6_kernel(const float* __restrict__ p, int deg, float* __restrict__ out) {
  const int nhb = head_blocks(deg);
  const int64_t total = x6_fwd_image_floats(deg);
  const int64_t bias_off = x6_fwd_bias_off(deg);
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    if (idx >= bias_off) {
      const int b = (int)(idx - bias_off);
      float v;
      if (b < 8 * kW) {
        v = p[leaf_bias_off(b / kW, deg) + (b % kW)];
      } else {
        const int n = b - 8 * kW, C = rgb_channels(deg);
        v = n < C ? p[leaf_bias_off(9, deg) + n] : (n == C ? p[leaf_bias_off(8, deg)] : 0.f);
      }
      out[idx] = v;
      continue;
    }
    int l = 0;
    while (l < 8 && idx >= x6_fwd_layer_off(l + 1)) ++l;
    const int64_t loc = idx - x6_fwd_layer_off(l);
    const int ncb = l < 8 ? 8 : nhb;
    const int s = (int)(loc & 3), lane = (int)((loc >> 2) & 63);
    const int64_t blk = loc >> 8;                        // (kg, cb, part)
    const int part = (int)(blk % 3);
    const int64_t kc = blk / 3;
    const int cb = (int)(kc % ncb), kg = (int)(kc / ncb);
    const int n = 32 * cb + (lane & 31);
    __bf16 pair[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int k = 16 * kg + 8 * (lane >> 5) + 2 * s + j;
      __bf16 a, b, c;
      split3(x6_src_weight(p, deg, l, k, n), a, b, c);
      pair[j] = part == 0 ? a : (part == 1 ? b : c);
    }
    uint32_t bits;
    __builtin_memcpy(&bits, pair, 4);
    reinterpret_cast<uint32_t*>(out)[idx] = bits;
  }
}

// backward(data): A[input feature n][k = output column] = W_l[n][k]; stream = head^T, then layers 7..1
__global__ void pack_bwd_x6_kernel(const float* __restrict__ p, int deg, float* __restrict__ out) {
  const int hk = x6_bwd_head_kg(deg);
  const int64_t total = x6_bwd_image_floats(deg);
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t kg_all = idx / kX6KgSlots;
    const int64_t loc = idx - kg_all * kX6KgSlots;
    int l, kg;
    if (kg_all < hk) { l = 8; kg = (int)kg_all; }
    else { l = 7 - (int)((kg_all - hk) / 16); kg = (int)((kg_all - hk) % 16); }
    const int s = (int)(loc & 3), lane = (int)((loc >> 2) & 63);
    const int blk = (int)(loc >> 8);                     // (cb, part)
    const int part = blk % 3, cb = blk / 3;
    const int n = 32 * cb + (lane & 31);
    __bf16 pair[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int k = 16 * kg + 8 * (lane >> 5) + 2 * s + j;
      __bf16 a, b, c;
      split3(x6_src_weight(p, deg, l, n, k), a, b, c);
      pair[j] = part == 0 ? a : (part == 1 ? b : c);
    }
    uint32_t bits;
    __builtin_memcpy(&bits, pair, 4);
    reinterpret_cast<uint32_t*>(out)[idx] = bits;
  }
}

int launch_pack_x6(const PxoCfg* cfg, const float* mlp_params, float* fwd, float* bwd, hipStream_t s) {
  hipLaunchKernelGGL(pack_fwd_x6_kernel, dim3(512), dim3(256), 0, s, mlp_params, cfg->sh_deg, fwd);
  if (bwd) hipLaunchKernelGGL(pack_bwd_x6_kernel, dim3(512), dim3(256), 0, s, mlp_params, cfg->sh_deg, bwd);
  return check_launch("pack_weights(bf16x6)");
}

// ------------------------------------------------------------------------------------------
// the sub-tile GEMM: acc[rb] (32 features x 32 samples, transposed) += W^T[features, K] X^T[K, samples of row block rb]
// ------------------------------------------------------------------------------------------
struct X6W { bf16x8 p[3]; };                  // the three parts of one (k-group, column block) weight fragment
template <int RBN> struct X6X { bf16x8 p[RBN][3]; };   // the three planes' fragments of RBN row blocks

// The packed image as a raw buffer (see mlp_kernels.hip make_wimage for the hardware assumption: reads past num_records
// return 0): the (layer, k-group, column block, part) part of a fragment's address is a wave-uniform byte offset in an SGPR,
// the lane part ONE 32-bit register that never changes.
struct X6Image {
  __amdgpu_buffer_rsrc_t rsrc;
  uint32_t voff;       // lane * 16
};
__device__ __forceinline__ X6Image make_x6image(const float* image, int64_t floats, int lane) {
  return X6Image{__builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(image), 0, (int)(floats * 4), 0x00020000), (uint32_t)lane * 16u};
}
// wu: wave-uniform f32x4 index of part 0 of this wave's fragment of k-group 0
__device__ __forceinline__ void load_w6(const X6Image& im, int wu, int kg, int kg_stride, X6W& w) {
  const int idx = wu + kg * kg_stride;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(im.rsrc, im.voff, (idx + 64 * i) * 16, 0));
    __builtin_memcpy(&w.p[i], &v, 16);
  }
}
// xp: this lane's row (lane & 31) and k offset 8 (lane >> 5) inside plane 0; planes are kPlane elements apart
template <int RBN>
__device__ __forceinline__ void load_x6(const __bf16* __restrict__ xp, int kg, X6X<RBN>& x) {
#pragma unroll
  for (int r = 0; r < RBN; ++r)
#pragma unroll
    for (int i = 0; i < 3; ++i) x.p[r][i] = *reinterpret_cast<const bf16x8*>(xp + i * kPlane + r * 32 * kLDB + kg * 16);
}

// hi: the leading product; lo: the five corrections, smallest first
template <int RBN>
__device__ __forceinline__ void mfma6(const X6W& w, const X6X<RBN>& x, f32x16 (&hi)[RBN], f32x16 (&lo)[RBN]) {
#define PXO_X6_MFMA(acc, wi, xi)                                                                      \
  _Pragma("unroll") for (int r = 0; r < RBN; ++r)                                                     \
      acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.p[wi], x.p[r][xi], acc[r], 0, 0, 0)
  PXO_X6_MFMA(hi, 0, 0);
  PXO_X6_MFMA(lo, 2, 0);
  PXO_X6_MFMA(lo, 0, 2);
  PXO_X6_MFMA(lo, 1, 1);
  PXO_X6_MFMA(lo, 1, 0);
  PXO_X6_MFMA(lo, 0, 1);
}
// the first k-group of a layer: the leading chain starts from `init` (the bias pattern: srcC is another register set, no
// copies), the correction chain from the inline constant 0
template <int RBN>
__device__ __forceinline__ void mfma6_first(const X6W& w, const X6X<RBN>& x, const f32x16& init, f32x16 (&hi)[RBN], f32x16 (&lo)[RBN]) {
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int r = 0; r < RBN; ++r) hi[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.p[0], x.p[r][0], init, 0, 0, 0);
#pragma unroll
  for (int r = 0; r < RBN; ++r) lo[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.p[2], x.p[r][0], zero, 0, 0, 0);
  PXO_X6_MFMA(lo, 0, 2);
  PXO_X6_MFMA(lo, 1, 1);
  PXO_X6_MFMA(lo, 1, 0);
  PXO_X6_MFMA(lo, 0, 1);
#undef PXO_X6_MFMA
}

// Weights (L2 latency) three k-groups ahead in four rotating register sets `w` (owned by the caller), activations (LDS) one
// ahead in two; kgroups must be a multiple of 4.  The weight fragments of a wave form ONE stream over the k-groups of
// consecutive layers (adjacent images of the same block shape), so the loads "past the end" of a layer fetch the first three
// k-groups of the next one: `avail` = k-groups that may be read starting at wu (>= kgroups), and a call with `preloaded` finds
// w[0..2] already holding its k-groups 0..2 -- the L2 latency of a layer's first fragments hides under the previous layer's
// tail and its epilogue.
#define PXO_X6_PIN() __builtin_amdgcn_sched_barrier(0)
__device__ __forceinline__ void lds_barrier6() {       // orders LDS traffic only (see mlp_kernels.hip lds_barrier)
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
// INIT: the accumulators are written, not read, by the first k-group (hi starts from *init, lo from 0)
template <int RBN, bool INIT = false>
__device__ __forceinline__ void gemm_x6(const __bf16* __restrict__ xp, const X6Image& im, int wu, int kgroups, int avail,
                                        bool preloaded, int kg_stride, X6W (&w)[4], f32x16 (&hi)[RBN], f32x16 (&lo)[RBN],
                                        const f32x16* init = nullptr) {
  X6X<RBN> x0, x1;
  const int last = avail - 1;
  auto cl = [&](int g) { return g < last ? g : last; };
  if (!preloaded) {
#pragma unroll
    for (int i = 0; i < 3; ++i) load_w6(im, wu, cl(i), kg_stride, w[i]);
  }
  load_x6<RBN>(xp, 0, x0);
  for (int g = 0; g < kgroups; g += 4) {
    load_x6<RBN>(xp, g + 1, x1);
    load_w6(im, wu, cl(g + 3), kg_stride, w[3]);
    PXO_X6_PIN();
    if (INIT && g == 0) mfma6_first<RBN>(w[0], x0, *init, hi, lo);
    else mfma6<RBN>(w[0], x0, hi, lo);
    PXO_X6_PIN();
    load_x6<RBN>(xp, g + 2, x0);
    load_w6(im, wu, cl(g + 4), kg_stride, w[0]);
    PXO_X6_PIN();
    mfma6<RBN>(w[1], x1, hi, lo);
    PXO_X6_PIN();
    load_x6<RBN>(xp, g + 3, x1);
    load_w6(im, wu, cl(g + 5), kg_stride, w[1]);
    PXO_X6_PIN();
    mfma6<RBN>(w[2], x0, hi, lo);
    PXO_X6_PIN();
    load_x6<RBN>(xp, (g + 4 < kgroups ? g + 4 : kgroups - 1), x0);
    load_w6(im, wu, cl(g + 6), kg_stride, w[2]);
    PXO_X6_PIN();
    mfma6<RBN>(w[3], x1, hi, lo);
    PXO_X6_PIN();
  }
}

// relu mask: one bit per accumulator element in (row block, quad, element) order, MSB first (mlp_kernels.hip mask_push / pop)
__device__ __forceinline__ void mask_push6(uint32_t& mw, float v) {
  asm volatile("v_cmp_lt_f32 vcc, 0, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(mw) : "v"(v) : "vcc");
}
// relu(s) and its mask bit from one compare: v = s > 0 ? s : 0, mw = 2 mw + (s > 0).  The condition lives in its own SGPR pair
// (VOP3 forms), not in VCC: consecutive elements do not serialise on one register and the compiler may interleave them.
__device__ __forceinline__ float relu_push6(uint32_t& mw, float s) {
  float v;
  unsigned long long c;
  asm("v_cmp_lt_f32 %1, 0, %3\n\tv_cndmask_b32 %0, 0, %3, %1\n\tv_addc_co_u32 %2, %1, %2, %2, %1"
      : "=&v"(v), "=&s"(c), "+v"(mw) : "v"(s));
  return v;
}
__device__ __forceinline__ float mask_pop6(uint32_t& mw, float x) {
  float r;
  unsigned long long c;
  asm("v_add_co_u32 %0, %2, %0, %0\n\tv_cndmask_b32 %1, 0, %3, %2" : "+v"(mw), "=&v"(r), "=&s"(c) : "v"(x));
  return r;
}

// the three planes' 8-byte pieces of four consecutive features of one sample, computed ahead of the barrier that frees the planes
struct X6Quad { uint2 p[3]; };
__device__ __forceinline__ X6Quad split_quad(f32x2 v01, f32x2 v23) {
  X6Quad o;
  split3_pair(v01, o.p[0].x, o.p[1].x, o.p[2].x);
  split3_pair(v23, o.p[0].y, o.p[1].y, o.p[2].y);
  return o;
}
// Written as they are (one ds_write_b64 per quad and plane) the 32 lanes of a half-wave store 8 bytes each at a row stride of
// 528 B: rows m and m + 16 fall on the same banks, and the 98 KB a layer writes take 1600 cycles with every matrix pipe idle
// (profiles/r06a_x6_phases.txt).  Quads are therefore paired (2p, 2p + 1) and v_permlane32_swap hands the lower half-wave both
// halves of quad 2p and the upper half-wave both halves of quad 2p + 1: every lane owns 8 CONSECUTIVE features and stores one
// 16-byte piece per plane -- the access pattern of the operand reads, which the row stride makes conflict-free.
struct X6Chunk { uint4 p[3]; };
__device__ __forceinline__ X6Chunk pair_quads(const X6Quad& a, const X6Quad& b) {
  X6Chunk o;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    // swap(a, b): [0] = {a.lower | b.lower}, [1] = {a.upper | b.upper} (lower / upper half-wave)
    const auto sx = __builtin_amdgcn_permlane32_swap(a.p[i].x, b.p[i].x, false, false);
    const auto sy = __builtin_amdgcn_permlane32_swap(a.p[i].y, b.p[i].y, false, false);
    o.p[i] = make_uint4(sx[0], sy[0], sx[1], sy[1]);
  }
  return o;
}
// e01 / e2: this lane's element offset in plane 0 / plane 2 (its own register: beyond the ds_write immediate range; an opaque
// INTEGER so that the access stays an LDS access -- an opaque pointer loses its address space and becomes a flat store)
__device__ __forceinline__ void store_chunk(__bf16* __restrict__ planes, int e01, int e2, int off, const X6Chunk& o) {
  *reinterpret_cast<uint4*>(planes + e01 + off) = o.p[0];
  *reinterpret_cast<uint4*>(planes + e01 + kPlane + off) = o.p[1];
  *reinterpret_cast<uint4*>(planes + e2 + off) = o.p[2];
}
__device__ __forceinline__ void store_planes4(__bf16* __restrict__ planes, int off, float v0, float v1, float v2, float v3) {
  uint32_t a01, b01, c01, a23, b23, c23;
  split3_pair(v0, v1, a01, b01, c01);
  split3_pair(v2, v3, a23, b23, c23);
  *reinterpret_cast<uint2*>(planes + off) = make_uint2(a01, a23);
  *reinterpret_cast<uint2*>(planes + kPlane + off) = make_uint2(b01, b23);
  *reinterpret_cast<uint2*>(planes + 2 * kPlane + off) = make_uint2(c01, c23);
}

// dense-grid point source (same formula as mlp_kernels.hip grid_point; octree/extraction.py:290-303)
struct X6Grid {
  int enabled, reso, x0;
  float off[3], scale[3];
};

__device__ __forceinline__ float x6_enc_value(float p0, float p1, float p2, int col) {
  if (col < 3) return col == 0 ? p0 : (col == 1 ? p1 : p2);
  if (col >= kEnc) return 0.f;
  int idx = col - 3;
  const bool shifted = idx >= 30;
  if (shifted) idx -= 30;
  const int l = idx / 3, a = idx - 3 * l;
  float xb = (a == 0 ? p0 : (a == 1 ? p1 : p2)) * (float)(1 << l);
  if (shifted) xb = xb + 1.5707963267948966f;
  return sinf(xb);
}

// posenc of the sub-tile's points, split, into planes[:, 0:64]: thread = (row = tid & 63, columns 8 wave .. 8 wave + 7 -- the
// column set is wave-uniform, so the branches of x6_enc_value are); `keep` receives the thread's three 16-byte pieces so that the
// skip layer can put them back without evaluating the sines again; SAVE: the float32 values leave for the weight gradients
template <bool SAVE, bool GRID>
__device__ __forceinline__ void posenc_tile_x6(__bf16* __restrict__ planes, const float* __restrict__ pts, const X6Grid& grid,
                                               int64_t row0, int64_t M, int tid, float* __restrict__ enc_out, uint4 (&keep)[3]) {
  const int row = tid & 63, part = tid >> 6;
  const int64_t grow = row0 + row;
  float p0 = 0.f, p1 = 0.f, p2 = 0.f;
  if (grow < M) {
    if (GRID && grid.enabled) {
      const int r = grid.reso;
      const int iz = (int)(grow % r);
      const int64_t t = grow / r;
      const int iy = (int)(t % r), ix = (int)(t / r) + grid.x0;
      p0 = ((((float)ix + 0.5f) / (float)r) - grid.off[0]) / grid.scale[0];
      p1 = ((((float)iy + 0.5f) / (float)r) - grid.off[1]) / grid.scale[1];
      p2 = ((((float)iz + 0.5f) / (float)r) - grid.off[2]) / grid.scale[2];
    } else {
      p0 = pts[grow * 3]; p1 = pts[grow * 3 + 1]; p2 = pts[grow * 3 + 2];
    }
  }
  float e[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) e[i] = x6_enc_value(p0, p1, p2, part * 8 + i);
  if (SAVE && grow < M) {
    float* dst = enc_out + grow * kEncPad + part * 8;
    *reinterpret_cast<f32x4*>(dst) = f32x4{e[0], e[1], e[2], e[3]};
    *reinterpret_cast<f32x4*>(dst + 4) = f32x4{e[4], e[5], e[6], e[7]};
  }
  uint32_t a[4], b[4], c[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) split3_pair(e[2 * i], e[2 * i + 1], a[i], b[i], c[i]);
  keep[0] = make_uint4(a[0], a[1], a[2], a[3]);
  keep[1] = make_uint4(b[0], b[1], b[2], b[3]);
  keep[2] = make_uint4(c[0], c[1], c[2], c[3]);
#pragma unroll
  for (int i = 0; i < 3; ++i) *reinterpret_cast<uint4*>(planes + i * kPlane + row * kLDB + part * 8) = keep[i];
}
__device__ __forceinline__ void posenc_restore_x6(__bf16* __restrict__ planes, int tid, const uint4 (&keep)[3]) {
  const int row = tid & 63, part = tid >> 6;
#pragma unroll
  for (int i = 0; i < 3; ++i) *reinterpret_cast<uint4*>(planes + i * kPlane + row * kLDB + part * 8) = keep[i];
}

// the persistent workgroups' tile schedule: identical to mlp_kernels.hip (TileTicket) -- slot b first, then slots taken from
// a device counter; results do not depend on it (a slot's rows, mask words and bias partial are functions of the slot alone)
struct X6Ticket {
  int* next;                       // LDS word
  unsigned int* counter;           // device word, zero when the launch starts
  __device__ __forceinline__ int draw(int tid) const { return tid == 0 ? (int)gridDim.x + (int)atomicAdd(counter, 1u) : 0; }
  __device__ __forceinline__ int64_t take(int tid, int ticket) const {
    if (tid == 0) *next = ticket;
    lds_barrier6();
    const int v = __builtin_amdgcn_readfirstlane(*next);
    lds_barrier6();
    return v;
  }
};

// ------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------
// One sub-tile of up to 64 rows starting at row0 through posenc, the 8 trunk layers and the heads.  mask_sub: this sub-tile's
// relu-mask words, [layer][thread] at stride 2 * kYThreads words per layer (SAVE only).
template <int NHB, bool SAVE, bool RGB, bool GRID>
__device__ __forceinline__ void fwd_subtile_x6(__bf16* __restrict__ planes, const float* __restrict__ s_bias,
                                               const float* __restrict__ pk, const float* __restrict__ pts, const X6Grid& grid,
                                               int64_t M, int deg, int64_t row0, float* __restrict__ raw_rgb,
                                               float* __restrict__ raw_sigma, float* __restrict__ acts,
                                               float* __restrict__ enc_out, uint32_t* __restrict__ mask_sub, int tid, int wave) {
  asm volatile("" : "+v"(tid));        // per-tile opaque thread id: addresses are derived inside the tile, not hoisted
  const int lane = tid & 63;
  const int C = rgb_channels(deg);
  const float* __restrict__ bias = pk + x6_fwd_bias_off(deg);
  const __bf16* xp = planes + (lane & 31) * kLDB + (lane >> 5) * 8;
  const X6Image wimg = make_x6image(pk, x6_fwd_image_floats(deg), lane);
  const int wu0 = wave * (3 * 64);                     // f32x4 index of this wave's column block inside a k-group
  constexpr int kTrunkKg = 4 + 16 * 4 + 20 + 16 * 2;   // 120 k-groups from layer 0 to layer 7
  constexpr int kKgStride = kX6KgSlots / 4;
  const int64_t rows = M - row0 < kYRows ? M - row0 : kYRows;
  // where this lane's quads go: element (row = 32 r + (lane & 31), feature = 32 wave + 8 q + 4 (lane >> 5)) -- one register each
  // for the row-major global copy and for planes 0-1 / plane 2 (beyond the ds_write immediate range); (r, q) are immediates
  const uint32_t st_voff = (uint32_t)((lane & 31) * kW + wave * 32 + 4 * (lane >> 5)) * 4u;
  const int pw = (lane & 31) * kLDB + wave * 32 + 8 * (lane >> 5);      // plane pieces: features 32 wave + 16 p + 8 (lane >> 5) .. + 7
  int pw2 = pw + 2 * kPlane;
  asm volatile("" : "+v"(pw2));

  lds_barrier6();   // the previous sub-tile's head GEMM has consumed the planes
  uint4 enc_keep[3];
  posenc_tile_x6<SAVE, GRID>(planes, pts, grid, row0, M, tid, enc_out, enc_keep);
  lds_barrier6();

  f32x16 hi[kYRB], lo[kYRB];
  X6W w[4];
  int kg0 = 0;          // position of the running layer in the wave's weight stream
  for (int l = 0; l < kDepth; ++l) {
    // the leading chain starts from the bias (register quad q of a lane holds features n0 .. n0 + 3 of one sample, whatever
    // the row block): the first k-group's MFMAs take it as srcC
    f32x16 binit;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(s_bias + l * kW + wave * 32 + 8 * q + 4 * (lane >> 5));
#pragma unroll
      for (int t = 0; t < 4; ++t) binit[4 * q + t] = b4[t];
    }
    const int nkg = l == 0 ? 4 : 16;
    gemm_x6<kYRB, true>(xp, wimg, wu0 + kg0 * kKgStride, nkg, kTrunkKg - kg0, l > 0, kKgStride, w, hi, lo, &binit);
    kg0 += nkg;
    if (l == 5) {
      // skip connection (model_utils.py:70-71): the 64 encoded columns are a second K segment, put back from registers
      lds_barrier6();
      posenc_restore_x6(planes, tid, enc_keep);
      lds_barrier6();
      gemm_x6<kYRB>(xp, wimg, wu0 + kg0 * kKgStride, 4, kTrunkKg - kg0, true, kKgStride, w, hi, lo);
      kg0 += 4;
    }
    if (l == kDepth - 1) {
      // the head slice's first weight fragments travel during this layer's epilogue (see the heads below)
      constexpr int NJ = RGB ? NHB : 1, NS = NHB == 3 ? 2 : 4, KS = 16 / NS;
      if (wave < NJ * NS) {
        const int wuh = (int)(x6_fwd_layer_off(8) / 4) + (RGB ? wave % NJ : NHB - 1) * (3 * 64) + (wave / NJ) * KS * (NHB * 3 * 64);
#pragma unroll
        for (int i = 0; i < 3; ++i) load_w6(wimg, wuh, i, NHB * 3 * 64, w[i]);
      }
    }
    // Epilogue, first half -- registers and global memory only, so it needs no barrier: a wave that is through its GEMM does
    // this vector work while slower waves still feed the matrix pipe.  relu + mask bit, the float32 copy for the weight
    // gradients, the exact three-way split.
    uint32_t mwr[kYRB] = {0u, 0u};                   // one mask chain per row block (16 bits each), joined below
    const __amdgpu_buffer_rsrc_t out =
        __builtin_amdgcn_make_buffer_rsrc(SAVE ? acts + ((int64_t)l * M + row0) * kW : nullptr, 0, SAVE ? (int)(rows * kW * 4) : 0, 0x00020000);
    X6Chunk pc[kYRB][2];
#pragma unroll
    for (int r = 0; r < kYRB; ++r) {
      X6Quad pq[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x2 s01 = f32x2{hi[r][4 * q], hi[r][4 * q + 1]} + f32x2{lo[r][4 * q], lo[r][4 * q + 1]};
        const f32x2 s23 = f32x2{hi[r][4 * q + 2], hi[r][4 * q + 3]} + f32x2{lo[r][4 * q + 2], lo[r][4 * q + 3]};
        f32x4 v;
        if (SAVE) {
          v[0] = relu_push6(mwr[r], s01[0]); v[1] = relu_push6(mwr[r], s01[1]);
          v[2] = relu_push6(mwr[r], s23[0]); v[3] = relu_push6(mwr[r], s23[1]);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), out, st_voff, (r * 32 * kW + 8 * q) * 4, 0);
        } else {
          v[0] = fmaxf(s01[0], 0.f); v[1] = fmaxf(s01[1], 0.f); v[2] = fmaxf(s23[0], 0.f); v[3] = fmaxf(s23[1], 0.f);
        }
        pq[q] = split_quad(f32x2{v[0], v[1]}, f32x2{v[2], v[3]});
      }
      pc[r][0] = pair_quads(pq[0], pq[1]);
      pc[r][1] = pair_quads(pq[2], pq[3]);
    }
    if (SAVE) mask_sub[(int64_t)l * (2 * kYThreads) + 2 * tid] = (mwr[0] << 16) | mwr[1];
    lds_barrier6();  // every wave has consumed the input planes
#pragma unroll
    for (int r = 0; r < kYRB; ++r)
#pragma unroll
      for (int p2 = 0; p2 < 2; ++p2) store_chunk(planes, pw, pw2, r * 32 * kLDB + 16 * p2, pc[r][p2]);
    lds_barrier6();
  }

  // heads (model_utils.py:72-74, :91-93): [raw_rgb | raw_sigma] = h7 @ [Dense_9 | Dense_8] + b.  The 32-column head blocks are too
  // few for 8 waves, and one block per wave with one row block per MFMA chain is bound by the L2 latency of its weight
  // fragments (12.8 k cycles per sub-tile, profiles/r06a_x6_phases.txt).  So K is cut into slices: wave w multiplies BOTH row
  // blocks by head block w % NJ over k-groups [KS (w / NJ), + KS) -- 12 MFMAs per weight fragment like the trunk, its first
  // fragments fetched before the last trunk epilogue -- the partial sums meet in LDS (the planes are free by then) and are
  // added in slice order, with the bias, by the thread that writes the output.
  {
    constexpr int NJ = RGB ? NHB : 1;                    // head blocks computed (sigma only: the block that holds column C)
    constexpr int NS = NHB == 3 ? 2 : 4;                 // K slices (by NHB, not NJ: sigma-only sums in the same order)
    constexpr int KS = 16 / NS;                          // k-groups per slice
    const bool active = wave < NJ * NS;
    const int cbj = wave % NJ, ks = wave / NJ;
    const int cb = RGB ? cbj : NHB - 1;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (active) {
      const int wuh = (int)(x6_fwd_layer_off(8) / 4) + cb * (3 * 64) + ks * KS * (NHB * 3 * 64);
      gemm_x6<kYRB, true>(xp + ks * KS * 16, wimg, wuh, KS, KS, true, NHB * 3 * 64, w, hi, lo, &zero16);
    }
    // partial sums in LDS as [slice][row][head column] (row stride RS floats), written as the 16-byte quads the lanes hold and
    // read back row-major: the sub-tile's rows of raw_rgb are ONE contiguous block of global memory, written coalesced
    constexpr int RS = NJ * 32 + 4;
    float* red = reinterpret_cast<float*>(planes);
    lds_barrier6();                                      // every wave is through with the planes
    if (active) {
#pragma unroll
      for (int r = 0; r < kYRB; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 v = {hi[r][4 * q] + lo[r][4 * q], hi[r][4 * q + 1] + lo[r][4 * q + 1], hi[r][4 * q + 2] + lo[r][4 * q + 2],
                           hi[r][4 * q + 3] + lo[r][4 * q + 3]};
          *reinterpret_cast<f32x4*>(red + (ks * kYRows + r * 32 + (lane & 31)) * RS + cbj * 32 + 8 * q + 4 * (lane >> 5)) = v;
        }
    }
    lds_barrier6();
    const float* hb = bias + 8 * kW;
    const int col0 = (RGB ? 0 : NHB - 1) * 32;           // head column of the first staged column
    if (RGB) {
      const int n_out = (int)rows * C;
      float* dst = raw_rgb + row0 * C;
      for (int o = tid; o < n_out; o += kYThreads) {
        const int row = o / C, col = o - row * C;
        float v = red[row * RS + col];
#pragma unroll
        for (int k = 1; k < NS; ++k) v += red[(k * kYRows + row) * RS + col];
        dst[o] = v + hb[col];
      }
    }
    if (tid < rows) {                                    // sigma: head column C
      float v = red[tid * RS + C - col0];
#pragma unroll
      for (int k = 1; k < NS; ++k) v += red[(k * kYRows + tid) * RS + C - col0];
      raw_sigma[row0 + tid] = v + hb[C];
    }
  }
}

template <int NHB, bool SAVE, bool RGB, bool GRID, bool DYN>
__global__ __launch_bounds__(kYThreads, 2) void mlp_fwd_x6_kernel(
    const float* __restrict__ pk, const float* __restrict__ pts, X6Grid grid, int64_t M, int deg, TileSched ts,
    float* __restrict__ raw_rgb, float* __restrict__ raw_sigma, float* __restrict__ acts, float* __restrict__ enc_out,
    uint32_t* __restrict__ mask, unsigned int* __restrict__ tile_counter) {
  __shared__ __attribute__((aligned(16))) __bf16 planes[3 * kPlane];
  __shared__ __attribute__((aligned(16))) float s_bias[kDepth * kW];      // trunk biases, staged once per workgroup
  __shared__ int s_next[4];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const float* __restrict__ bias = pk + x6_fwd_bias_off(deg);
  for (int i = tid; i < kDepth * kW; i += kYThreads) s_bias[i] = bias[i];
  auto run = [&](int64_t slot) {
    uint32_t* mslot = SAVE ? mask + slot * kDepth * (2 * kYThreads) : nullptr;
    if (slot < ts.n_full) {
#pragma unroll 1
      for (int sub = 0; sub < 2; ++sub) {
        const int64_t row0 = slot * kTM + sub * kYRows;
        if (row0 < M)
          fwd_subtile_x6<NHB, SAVE, RGB, GRID>(planes, s_bias, pk, pts, grid, M, deg, row0, raw_rgb, raw_sigma, acts, enc_out,
                                               mslot + sub, tid, wave);
      }
    } else {
      fwd_subtile_x6<NHB, SAVE, RGB, GRID>(planes, s_bias, pk, pts, grid, M, deg, ts.half_row0 + (slot - ts.n_full) * kYRows,
                                           raw_rgb, raw_sigma, acts, enc_out, mslot, tid, wave);
    }
  };
  const int64_t n_slots = ts.n_full + ts.n_half;
  if (!DYN) {
    for (int64_t slot = blockIdx.x; slot < n_slots; slot += gridDim.x) run(slot);
  } else {
    const X6Ticket tk{s_next, tile_counter};
    for (int64_t slot = blockIdx.x; slot < n_slots;) {
      const int ticket = tk.draw(tid);
      run(slot);
      slot = tk.take(tid, ticket);
    }
  }
}

static unsigned x6_grid(int64_t M) {
  const int64_t tiles = num_tiles(M), cap = (int64_t)num_cus();
  return (unsigned)(tiles < cap ? tiles : cap);
}

template <int NHB>
static int launch_fwd_x6_nhb(const PxoCfg* cfg, const float* pk, const float* pts, const X6Grid& grid, int64_t M,
                             float* raw_rgb, float* raw_sigma, float* acts, float* enc, uint32_t* mask,
                             unsigned int* tile_counter, hipStream_t s) {
  KernelTimer timer(PXO_PROF_MLP_FWD, M, s);
  dim3 grid_dim(x6_grid(M)), block(kYThreads);
  const TileSched ts = tile_sched(M, grid_dim.x);
#define PXO_X6_FWD(SAVE_, RGB_, GRID_, DYN_)                                                                               \
  hipLaunchKernelGGL((mlp_fwd_x6_kernel<NHB, SAVE_, RGB_, GRID_, DYN_>), grid_dim, block, 0, s, pk, pts, grid, M, cfg->sh_deg, \
                     ts, raw_rgb, raw_sigma, acts, enc, mask, tile_counter)
  if (acts && raw_rgb && tile_counter) PXO_X6_FWD(true, true, false, true);
  else if (acts && raw_rgb) PXO_X6_FWD(true, true, false, false);
  else if (acts) PXO_X6_FWD(true, false, false, false);
  else if (grid.enabled) PXO_X6_FWD(false, false, true, false);
  else if (raw_rgb) PXO_X6_FWD(false, true, false, false);
  else PXO_X6_FWD(false, false, false, false);
#undef PXO_X6_FWD
  return check_launch("mlp_fwd(bf16x6)");
}

int launch_mlp_fwd_x6(const PxoCfg* cfg, const float* packed_fwd, const float* pts, int reso, int x0, const float* off,
                      const float* scale, int64_t M, float* raw_rgb, float* raw_sigma, float* acts, float* enc,
                      uint32_t* mask, unsigned int* tile_counter, hipStream_t s) {
  if (M == 0) return PXO_OK;
  X6Grid g;
  g.enabled = pts == nullptr; g.reso = reso > 0 ? reso : 1; g.x0 = x0;
  for (int i = 0; i < 3; ++i) { g.off[i] = off ? off[i] : 0.f; g.scale[i] = scale ? scale[i] : 1.f; }
  if (g.enabled && (raw_rgb || acts)) { set_error("mlp_fwd(bf16x6): the dense-grid source is sigma-only"); return PXO_ERR_ARG; }
  switch (head_blocks(cfg->sh_deg)) {
    case 1: return launch_fwd_x6_nhb<1>(cfg, packed_fwd, pts, g, M, raw_rgb, raw_sigma, acts, enc, mask, tile_counter, s);
    case 2: return launch_fwd_x6_nhb<2>(cfg, packed_fwd, pts, g, M, raw_rgb, raw_sigma, acts, enc, mask, tile_counter, s);
    default: return launch_fwd_x6_nhb<3>(cfg, packed_fwd, pts, g, M, raw_rgb, raw_sigma, acts, enc, mask, tile_counter, s);
  }
}

// ------------------------------------------------------------------------------------------
// backward (data): d_raw -> dz_7 .. dz_0
// ------------------------------------------------------------------------------------------
// sum over the 32 lanes that share (lane >> 5), in a fixed order; the total lands in lanes 16..31 of each half
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float sum_half_wave(float v) {
  v = dpp_add<0xB1, 0xf>(v);     // quad_perm [1,0,3,2]
  v = dpp_add<0x4E, 0xf>(v);     // quad_perm [2,3,0,1]
  v = dpp_add<0x141, 0xf>(v);    // row_half_mirror
  v = dpp_add<0x140, 0xf>(v);    // row_mirror: every lane of a 16-lane row holds the row's sum
  v = dpp_add<0x142, 0xa>(v);    // row_bcast15 into rows 1 and 3: lanes 16..31 / 48..63 hold the 32-lane sums
  return v;
}

// One sub-tile of up to 64 rows.  db_acc: the slot's [9][256] bias partial in LDS (accumulate = false: this sub-tile writes it,
// true: adds to it).  Returns false when the sub-tile was skipped (SKIP and no live row).
template <int NHB, bool SKIP>
__device__ __forceinline__ bool bwd_subtile_x6(__bf16* __restrict__ planes, float* __restrict__ stage, float* __restrict__ db_acc,
                                               int* __restrict__ nz, const float* __restrict__ pkb,
                                               const float* __restrict__ d_raw_rgb, const float* __restrict__ d_raw_sigma,
                                               const uint32_t* __restrict__ mask_sub, int64_t M, int deg, int64_t row0,
                                               float* __restrict__ dz, uint8_t* __restrict__ chunk_live, bool accumulate,
                                               int tid, int wave, bool db_all) {
  constexpr int NH = 32 * NHB;                           // head columns (C rgb + sigma + zero padding)
  constexpr int HK = 4 * ((NHB + 1) / 2);                // k-groups of the head^T GEMM (zero-padded)
  constexpr int NHP = 16 * HK;                           // staged columns
  constexpr int kLDS = NHP + 4;                          // staging row stride (floats)
  constexpr int kChunks = kYRows / kLiveRows;            // 4
  constexpr int kKgStride = kX6KgSlots / 4;
  constexpr int kStreamKg = HK + 7 * 16;
  asm volatile("" : "+v"(tid));
  const int lane = tid & 63;
  const int C = rgb_channels(deg);
  const __bf16* xp = planes + (lane & 31) * kLDB + (lane >> 5) * 8;
  const X6Image wimg = make_x6image(pkb, x6_bwd_image_floats(deg), lane);
  const int wu0 = wave * (3 * 64);
  const int64_t rows = M - row0 < kYRows ? M - row0 : kYRows;
  const uint32_t st_voff = (uint32_t)((lane & 31) * kW + wave * 32 + 4 * (lane >> 5)) * 4u;     // see fwd_subtile_x6
  const int pw = (lane & 31) * kLDB + wave * 32 + 8 * (lane >> 5);      // plane pieces: features 32 wave + 16 p + 8 (lane >> 5) .. + 7
  int pw2 = pw + 2 * kPlane;
  asm volatile("" : "+v"(pw2));
  // this lane's bias accumulators: lane 31 / 63 of a wave writes the sums of features 32 wave + 4 (lane >> 5) + 8 q + t
  float* dbp = db_acc + wave * 32 + 4 * (lane >> 5);

  lds_barrier6();   // the previous sub-tile is through with the planes, the staging tile and nz
  if (SKIP && tid < kChunks) nz[tid] = 0;
  if (SKIP) lds_barrier6();
  // d_raw sub-tile -> stage[:, 0:NHP] with the head's column order (d_raw_rgb == NULL: sigma-only rows)
  for (int idx = tid; idx < kYRows * NHP; idx += kYThreads) {
    const int row = idx / NHP, col = idx - row * NHP;
    const int64_t grow = row0 + row;
    float v = 0.f;
    if (grow < M) {
      if (col < C) { if (d_raw_rgb) v = d_raw_rgb[grow * C + col]; }
      else if (col == C) v = d_raw_sigma[grow];
    }
    stage[row * kLDS + col] = v;
    if (SKIP && v != 0.f) nz[row / kLiveRows] = 1;                // same value from every writer
  }
  lds_barrier6();
  if (SKIP) {
    // see mlp_kernels.hip bwd_tile: rows with an exactly zero upstream gradient add exactly nothing anywhere
    int any = 0;
#pragma unroll
    for (int c = 0; c < kChunks; ++c) any |= nz[c];
    if (tid < kChunks) chunk_live[row0 / kLiveRows + tid] = (uint8_t)nz[tid];
    if (!any) return false;
  }
  if (tid < kW) {  // head bias gradient: this sub-tile's column sums
    float sum = 0.f;
    if (tid < NH) {
#pragma unroll 8
      for (int row = 0; row < kYRows; ++row) sum += stage[row * kLDS + tid];
    }
    db_acc[8 * kW + tid] = accumulate ? db_acc[8 * kW + tid] + sum : sum;
  }
  // split the staged tile into the planes: item = (row, group of 8 columns)
  for (int item = tid; item < kYRows * (NHP / 8); item += kYThreads) {
    const int row = item & 63, grp = item >> 6;
    const f32x4 u0 = *reinterpret_cast<const f32x4*>(stage + row * kLDS + grp * 8);
    const f32x4 u1 = *reinterpret_cast<const f32x4*>(stage + row * kLDS + grp * 8 + 4);
    uint32_t a[4], b[4], c[4];
    split3_pair(u0[0], u0[1], a[0], b[0], c[0]);
    split3_pair(u0[2], u0[3], a[1], b[1], c[1]);
    split3_pair(u1[0], u1[1], a[2], b[2], c[2]);
    split3_pair(u1[2], u1[3], a[3], b[3], c[3]);
    *reinterpret_cast<uint4*>(planes + row * kLDB + grp * 8) = make_uint4(a[0], a[1], a[2], a[3]);
    *reinterpret_cast<uint4*>(planes + kPlane + row * kLDB + grp * 8) = make_uint4(b[0], b[1], b[2], b[3]);
    *reinterpret_cast<uint4*>(planes + 2 * kPlane + row * kLDB + grp * 8) = make_uint4(c[0], c[1], c[2], c[3]);
  }
  lds_barrier6();

  f32x16 hi[kYRB], lo[kYRB];
  X6W w[4];
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  uint32_t mw = mask_sub[(int64_t)(kDepth - 1) * (2 * kYThreads) + 2 * tid];   // relu mask of the layer the running GEMM produces
  int kg0 = 0;
  gemm_x6<kYRB, true>(xp, wimg, wu0, HK, kStreamKg, false, kKgStride, w, hi, lo, &zero16);
  kg0 += HK;
  for (int l = kDepth - 1; l >= 0; --l) {
    // Epilogue, first half -- registers, global memory and this lane's own bias accumulators only: no barrier needed, so a
    // wave that is through its GEMM does this vector work while the other wave of its SIMD still feeds the matrix pipe
    const __amdgpu_buffer_rsrc_t out =
        __builtin_amdgcn_make_buffer_rsrc(dz + ((int64_t)l * M + row0) * kW, 0, (int)(rows * kW * 4), 0x00020000);
    f32x2 cs[8];
    X6Chunk pc[kYRB][2];
    uint32_t mwr[kYRB] = {mw, mw << 16};             // one mask chain per row block (MSB first)
#pragma unroll
    for (int r = 0; r < kYRB; ++r) {
      X6Quad pq[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x2 s01 = f32x2{hi[r][4 * q], hi[r][4 * q + 1]} + f32x2{lo[r][4 * q], lo[r][4 * q + 1]};
        const f32x2 s23 = f32x2{hi[r][4 * q + 2], hi[r][4 * q + 3]} + f32x2{lo[r][4 * q + 2], lo[r][4 * q + 3]};
        f32x2 v01, v23;
        v01[0] = mask_pop6(mwr[r], s01[0]); v01[1] = mask_pop6(mwr[r], s01[1]);
        v23[0] = mask_pop6(mwr[r], s23[0]); v23[1] = mask_pop6(mwr[r], s23[1]);
        cs[2 * q] = r == 0 ? v01 : cs[2 * q] + v01;
        cs[2 * q + 1] = r == 0 ? v23 : cs[2 * q + 1] + v23;
        const f32x4 v = {v01[0], v01[1], v23[0], v23[1]};
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), out, st_voff, (r * 32 * kW + 8 * q) * 4, 0);
        if (l > 0) pq[q] = split_quad(v01, v23);
      }
      if (l > 0) {
        pc[r][0] = pair_quads(pq[0], pq[1]);
        pc[r][1] = pair_quads(pq[2], pq[3]);
      }
    }
    // bias gradient of layer l: column sums over the sub-tile's samples (= lanes), fixed order.  !db_all: Dense_1..7's come
    // from the weight-gradient kernel, which sums the columns of dz_1..7 while it streams them (pxo_common.h kBiasFromWgrad)
    if (db_all || l == 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float sum = sum_half_wave(cs[i >> 1][i & 1]);
        if ((lane & 31) == 31) {
          float* p = dbp + l * kW + 8 * (i >> 2) + (i & 3);
          *p = accumulate ? *p + sum : sum;
        }
      }
    }
    if (l > 0) mw = mask_sub[(int64_t)(l - 1) * (2 * kYThreads) + 2 * tid];     // next layer's mask, fetched under the GEMM
    lds_barrier6();  // every wave has consumed the planes (and, at l == 0, the slot's bias sums are complete)
    if (l == 0) break;
#pragma unroll
    for (int r = 0; r < kYRB; ++r)
#pragma unroll
      for (int p2 = 0; p2 < 2; ++p2) store_chunk(planes, pw, pw2, r * 32 * kLDB + 16 * p2, pc[r][p2]);
    lds_barrier6();
    gemm_x6<kYRB, true>(xp, wimg, wu0 + kg0 * kKgStride, 16, kStreamKg - kg0, true, kKgStride, w, hi, lo, &zero16);
    kg0 += 16;
  }
  return true;
}

template <int NHB, bool SKIP, bool DYN>
__global__ __launch_bounds__(kYThreads, 2) void mlp_bwd_data_x6_kernel(
    const float* __restrict__ pkb, const float* __restrict__ d_raw_rgb, const float* __restrict__ d_raw_sigma,
    const uint32_t* __restrict__ mask, int64_t M, int deg, TileSched ts, float* __restrict__ dz,
    float* __restrict__ dbias_partial, uint8_t* __restrict__ chunk_live, unsigned int* __restrict__ tile_counter, int db_all) {
  constexpr int HK = 4 * ((NHB + 1) / 2);
  __shared__ __attribute__((aligned(16))) __bf16 planes[3 * kPlane];
  __shared__ __attribute__((aligned(16))) float stage[kYRows * (16 * HK + 4)];
  __shared__ __attribute__((aligned(16))) float db_acc[9 * kW];
  __shared__ int nz[8];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  uint8_t* const tile_live = reinterpret_cast<uint8_t*>(dbias_partial + mask_slots(M) * 9 * kW);
  auto run = [&](int64_t slot) {
    const uint32_t* mslot = mask + slot * kDepth * (2 * kYThreads);
    bool any = false;
    if (slot < ts.n_full) {
#pragma unroll 1
      for (int sub = 0; sub < 2; ++sub) {
        const int64_t row0 = slot * kTM + sub * kYRows;
        if (row0 < M)
          any |= bwd_subtile_x6<NHB, SKIP>(planes, stage, db_acc, nz, pkb, d_raw_rgb, d_raw_sigma, mslot + sub, M, deg, row0, dz,
                                           chunk_live, any, tid, wave, db_all != 0);
        else if (SKIP && tid < kYRows / kLiveRows)
          chunk_live[row0 / kLiveRows + tid] = 0;
      }
    } else {
      any = bwd_subtile_x6<NHB, SKIP>(planes, stage, db_acc, nz, pkb, d_raw_rgb, d_raw_sigma, mslot, M, deg,
                                      ts.half_row0 + (slot - ts.n_full) * kYRows, dz, chunk_live, false, tid, wave, db_all != 0);
    }
    if (tid == 0) tile_live[slot] = (uint8_t)(any ? 1 : 0);
    if (any) {       // the slot's bias partial leaves LDS (the last epilogue ended with a barrier)
      float* db = dbias_partial + slot * 9 * kW;
      for (int i = tid; i < 9 * kW; i += kYThreads)
        if (db_all || i < kW || i >= 8 * kW) db[i] = db_acc[i];         // !db_all: rows 1..7 were not computed and are not read
    }
  };
  const int64_t n_slots = ts.n_full + ts.n_half;
  if (!DYN) {
    for (int64_t slot = blockIdx.x; slot < n_slots; slot += gridDim.x) run(slot);
  } else {
    const X6Ticket tk{nz + 4, tile_counter};
    for (int64_t slot = blockIdx.x; slot < n_slots;) {
      const int ticket = tk.draw(tid);
      run(slot);
      slot = tk.take(tid, ticket);
    }
  }
}

int launch_mlp_bwd_data_x6(const PxoCfg* cfg, const float* packed_bwd, const float* d_raw_rgb, const float* d_raw_sigma,
                           const uint32_t* mask, int64_t M, float* dz, float* dbias_partial, uint8_t* chunk_live,
                           unsigned int* tile_counter, hipStream_t s, bool db_all) {
  if (M == 0) return PXO_OK;
  KernelTimer timer(PXO_PROF_MLP_BWD_DATA, M, s);
  dim3 grid_dim(x6_grid(M)), block(kYThreads);
  const TileSched ts = tile_sched(M, grid_dim.x);
#define PXO_X6_BWD_(NHB_, SKIP_, DYN_)                                                                                        \
  hipLaunchKernelGGL((mlp_bwd_data_x6_kernel<NHB_, SKIP_, DYN_>), grid_dim, block, 0, s, packed_bwd, d_raw_rgb, d_raw_sigma, \
                     mask, M, cfg->sh_deg, ts, dz, dbias_partial, chunk_live, tile_counter, db_all ? 1 : 0)
#define PXO_X6_BWD(NHB_)                                                    \
  do {                                                                      \
    if (chunk_live && tile_counter) PXO_X6_BWD_(NHB_, true, true);          \
    else if (chunk_live) PXO_X6_BWD_(NHB_, true, false);                    \
    else if (tile_counter) PXO_X6_BWD_(NHB_, false, true);                  \
    else PXO_X6_BWD_(NHB_, false, false);                                   \
  } while (0)
  switch (head_blocks(cfg->sh_deg)) {
    case 1: PXO_X6_BWD(1); break;
    case 2: PXO_X6_BWD(2); break;
    default: PXO_X6_BWD(3); break;
  }
#undef PXO_X6_BWD
#undef PXO_X6_BWD_
  return check_launch("mlp_bwd_data(bf16x6)");
}

}  // namespace pxo
