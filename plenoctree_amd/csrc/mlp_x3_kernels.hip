// Opt-in split-precision forward of the NeRF-SH MLP for gfx950: PxoCfg.mlp_precision = PXO_MLP_BF16X3.
//
// INFERENCE ONLY (pxo_eval_points / pxo_grid_sigma / pxo_render_fwd / pxo_mlp_fwd without saved tensors): the
// training path and every headline number stay on the exact-f32 kernels of mlp_kernels.hip.
//
// Every f32 operand x is split as x = hi + lo + O(2^-17 |x|), hi = bf16(x), lo = bf16(x - hi) (round to nearest
// even, v_cvt_pk_bf16_f32), and a product is evaluated as  hi_a hi_b + hi_a lo_b + lo_a hi_b  -- three
// v_mfma_f32_32x32x16_bf16 with float32 accumulation (bf16 x bf16 products are exact in f32; the dropped lo_a lo_b term
// is 2^-16 relative).  The bf16 MFMA runs at 16x the f32 MFMA rate, so the three passes cost 3/16 of the f32 GEMM.
// Measured against the float64 oracle the rendered colours differ by |dPSNR| ~ 2e-5 dB (the bar of north_star is
// 1e-4 dB; tests/test_gpu_x3.py), i.e. as close to float64 as the float32 evaluation is.
//
// Replaces posenc + MLP.__call__ (nerf_sh/nerf/model_utils.py:43-94,145-173; torch twin
// octree/nerf/model_utils.py:87-158) for the same configurations as mlp_fwd_kernel.
//
// Data flow (same persistent-tile structure as mlp_fwd_kernel, different operand plumbing):
//  * the 128-row activation tile lives in LDS as TWO bf16 planes (hi, lo), row stride 264 bf16 = 528 B (conflict-free
//    ds_read_b128 / ds_write_b64); same 132 KB footprint as the f32 tile;
//  * weights are pre-split and packed per (16-wide k-group, 32-column block) as [hi | lo] fragments of 64 lanes x
//    16 B, read straight from L2 into registers three k-groups ahead (image size and layer offsets are those of
//    the f32 image: 4 bytes per weight either way);
//  * the product is computed TRANSPOSED (weights as the A operand, activations as B): a lane then owns 4 consecutive
//    output features of one sample per register quad, so the epilogue (bias, ReLU, split) packs 4 bf16 and writes
//    one ds_write_b64 per plane instead of 8 scattered 2-byte writes.
#include "pxo_common.h"

namespace pxo {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int kLDB = 264;                    // LDS row stride in bf16 (256 + 8: one b128 access of pad)

__device__ __forceinline__ void split_bf16(float x, __bf16& hi, __bf16& lo) {
  hi = (__bf16)x;
  lo = (__bf16)(x - (float)hi);
}
// the same for a pair, packed: one v_cvt_pk_bf16_f32 per half, hi widened back with a shift / a mask
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_bf16_pair(float a, float b, uint32_t& hi, uint32_t& lo) {
  const bf16x2 h = {(__bf16)a, (__bf16)b};
  uint32_t hb;
  __builtin_memcpy(&hb, &h, 4);
  const float ha = __uint_as_float(hb << 16), hbf = __uint_as_float(hb & 0xffff0000u);
  const bf16x2 l = {(__bf16)(a - ha), (__bf16)(b - hbf)};
  __builtin_memcpy(&lo, &l, 4);
  hi = hb;
}

// ------------------------------------------------------------------------------------------
// weight packing: element e of lane l in block (kg, cb, part) = part(W[k = 16 kg + 8 (l >> 5) + e][n = 32 cb + (l & 31)])
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float x3_src_weight(const float* __restrict__ p, int deg, int l, int k, int n) {
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

__global__ void pack_fwd_x3_kernel(const float* __restrict__ p, int deg, float* __restrict__ out) {
  const int nhb = head_blocks(deg);
  const int64_t total = fwd_image_floats(deg);
  const int64_t bias_off = fwd_bias_off(deg);
  // one thread per 4-byte slot: below bias_off a slot holds two bf16 (elements 2s, 2s+1 of a 16-byte lane fragment)
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
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
    while (l < 8 && idx >= fwd_layer_off(l + 1)) ++l;
    const int64_t loc = idx - fwd_layer_off(l);          // in 4-byte slots
    const int ncb = l < 8 ? 8 : nhb;
    const int s = (int)(loc & 3), lane = (int)((loc >> 2) & 63);
    const int64_t blk = loc >> 8;                        // (kg, cb, part): 64 lanes x 4 slots = 256 slots per block
    const int part = (int)(blk & 1);
    const int64_t kc = blk >> 1;
    const int cb = (int)(kc % ncb), kg = (int)(kc / ncb);
    const int n = 32 * cb + (lane & 31);
    __bf16 pair[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int k = 16 * kg + 8 * (lane >> 5) + 2 * s + j;
      __bf16 hi, lo;
      split_bf16(x3_src_weight(p, deg, l, k, n), hi, lo);
      pair[j] = part == 0 ? hi : lo;
    }
    uint32_t bits;
    __builtin_memcpy(&bits, pair, 4);
    reinterpret_cast<uint32_t*>(out)[idx] = bits;
  }
}

int launch_pack_x3(const PxoCfg* cfg, const float* mlp_params, float* fwd, hipStream_t s) {
  hipLaunchKernelGGL(pack_fwd_x3_kernel, dim3(512), dim3(256), 0, s, mlp_params, cfg->sh_deg, fwd);
  return check_launch("pack_weights(bf16x3)");
}

// ------------------------------------------------------------------------------------------
// the tile GEMM: acc[rb] (32 features x 32 samples, transposed) += W^T[features, K] X^T[K, samples of row block rb]
// ------------------------------------------------------------------------------------------
struct X3Frag { bf16x8 hi, lo; };

// Geometry (both were built and measured; 1 is what ships):
//   0: 128-row tiles, one 8-wave workgroup per CU, every wave 128 rows x 32 features (the f32 kernel's geometry)
//   1: 64-row tiles, TWO independent 4-wave workgroups per CU, every wave 64 rows x 64 features.  Per MFMA it reads half
//      as many activation fragments from LDS and twice as many weight fragments from L2; the two workgroups drift
//      apart, so one's posenc / epilogue / prologue phases run under the other's MFMAs.  With the three bf16 passes
//      costing 3/16 of the f32 GEMM those phases are no longer small against the GEMM, which is why the geometry that
//      lost for f32 (138 vs 150 TFLOP/s in the bare-loop probe) is measured here.
constexpr int kXRows = 64, kXWaves = 4, kXCB = 2, kXWgPerCu = 2;
constexpr int kXThreads = kXWaves * 64;
constexpr int kXRB = kXRows / 32;

// The packed image as a raw buffer (see mlp_kernels.hip make_wimage for the hardware assumption: reads past num_records
// return 0): the (layer, k-group, column block, part) part of a fragment's address is a wave-uniform byte offset in an SGPR, the
// lane part ONE 32-bit register that never changes -- no 64-bit vector address arithmetic per load, no per-lane pointers held
// in registers across the GEMMs (round 5: the kernel sat at the 256-VGPR cap with up to 196 B per lane of scratch spills).
struct X3Image {
  __amdgpu_buffer_rsrc_t rsrc;
  uint32_t voff;       // lane * 16
};
__device__ __forceinline__ X3Image make_x3image(const float* image, int64_t floats, int lane) {
  return X3Image{__builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(image), 0, (int)(floats * 4), 0x00020000), (uint32_t)lane * 16u};
}
// wu: wave-uniform f32x4 index of the first fragment of k-group 0 (of this wave's column blocks) inside the image
template <int CBN>
__device__ __forceinline__ void load_w(const X3Image& im, int wu, int kg, int kg_stride, X3Frag (&w)[CBN]) {
#pragma unroll
  for (int c = 0; c < CBN; ++c) {
    const int idx = wu + kg * kg_stride + c * 128;
    const f32x4 h = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(im.rsrc, im.voff, idx * 16, 0));
    const f32x4 l = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(im.rsrc, im.voff, (idx + 64) * 16, 0));
    __builtin_memcpy(&w[c].hi, &h, 16);
    __builtin_memcpy(&w[c].lo, &l, 16);
  }
}

template <int RBN>
__device__ __forceinline__ void load_x(const __bf16* __restrict__ xh, const __bf16* __restrict__ xl, int kg, X3Frag (&x)[RBN]) {
#pragma unroll
  for (int r = 0; r < RBN; ++r) {
    x[r].hi = *reinterpret_cast<const bf16x8*>(xh + r * 32 * kLDB + kg * 16);
    x[r].lo = *reinterpret_cast<const bf16x8*>(xl + r * 32 * kLDB + kg * 16);
  }
}

template <int RBN, int CBN>
__device__ __forceinline__ void mfma3(const X3Frag (&w)[CBN], const X3Frag (&x)[RBN], f32x16 (&acc)[RBN][CBN]) {
#pragma unroll
  for (int r = 0; r < RBN; ++r)
#pragma unroll
    for (int c = 0; c < CBN; ++c) acc[r][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[c].hi, x[r].hi, acc[r][c], 0, 0, 0);
#pragma unroll
  for (int r = 0; r < RBN; ++r)
#pragma unroll
    for (int c = 0; c < CBN; ++c) acc[r][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[c].hi, x[r].lo, acc[r][c], 0, 0, 0);
#pragma unroll
  for (int r = 0; r < RBN; ++r)
#pragma unroll
    for (int c = 0; c < CBN; ++c) acc[r][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[c].lo, x[r].hi, acc[r][c], 0, 0, 0);
}

// Weights (L2 latency) three k-groups ahead in four rotating register sets `w` (owned by the caller), activations
// (LDS) one ahead in two; kgroups must be a multiple of 4.  The weight fragments of a wave form ONE stream over the
// k-groups of consecutive trunk layers (the images of layers l and l+1 are adjacent and have the same block shape), so
// the loads "past the end" of a layer fetch the first three k-groups of the next one: `avail` = k-groups that may be
// read starting at wp (>= kgroups), and a call with `preloaded` finds w[0..2] already holding its k-groups 0..2 --
// the L2 latency of a layer's first fragments is then hidden under the previous layer's tail and its epilogue.
// xh / xl: this lane's row (lane & 31) and k offset 8 (lane >> 5) inside the planes.
#define PXO_X3_PIN() __builtin_amdgcn_sched_barrier(0)
template <int RBN, int CBN>
__device__ __forceinline__ void gemm_x3(const __bf16* __restrict__ xh, const __bf16* __restrict__ xl,
                                        const X3Image& im, int wu, int kgroups, int avail, bool preloaded, int kg_stride,
                                        X3Frag (&w)[4][CBN], f32x16 (&acc)[RBN][CBN]) {
  X3Frag x0[RBN], x1[RBN];
  const int last = avail - 1;
  auto cl = [&](int g) { return g < last ? g : last; };
  if (!preloaded) {
#pragma unroll
    for (int i = 0; i < 3; ++i) load_w<CBN>(im, wu, cl(i), kg_stride, w[i]);
  }
#define LOADX(g, x) load_x<RBN>(xh, xl, g, x)
#define LOADW(g, ww) load_w<CBN>(im, wu, cl(g), kg_stride, ww)
  load_x<RBN>(xh, xl, 0, x0);
  for (int g = 0; g < kgroups; g += 4) {
    LOADX(g + 1, x1);
    LOADW(g + 3, w[3]);
    PXO_X3_PIN();
    mfma3<RBN, CBN>(w[0], x0, acc);
    PXO_X3_PIN();
    LOADX(g + 2, x0);
    LOADW(g + 4, w[0]);
    PXO_X3_PIN();
    mfma3<RBN, CBN>(w[1], x1, acc);
    PXO_X3_PIN();
    LOADX(g + 3, x1);
    LOADW(g + 5, w[1]);
    PXO_X3_PIN();
    mfma3<RBN, CBN>(w[2], x0, acc);
    PXO_X3_PIN();
    LOADX((g + 4 < kgroups ? g + 4 : kgroups - 1), x0);
    LOADW(g + 6, w[2]);
    PXO_X3_PIN();
    mfma3<RBN, CBN>(w[3], x1, acc);
    PXO_X3_PIN();
  }
}

// dense-grid point source (same formula as mlp_kernels.hip; octree/extraction.py:290-303)
struct X3Grid {
  int enabled, reso, x0;
  float off[3], scale[3];
};

__device__ __forceinline__ float x3_enc_value(float p0, float p1, float p2, int col) {
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

// posenc of the tile's points, split, into planes[:, 0:64]; `keep` receives this thread's four 16-byte pieces (hi, lo of its
// two 8-column halves) so that the skip layer can put them back (posenc_restore_x3) instead of evaluating 16 sinf per point a
// second time
__device__ __forceinline__ void posenc_tile_x3(__bf16* __restrict__ ph, __bf16* __restrict__ pl, const float* __restrict__ pts,
                                               const X3Grid& grid, int64_t row0, int64_t M, int tid, uint4 (&keep)[4]) {
  static_assert(kXThreads / kXRows == 4, "4 parts x 16 columns");
  const int row = tid % kXRows, part = tid / kXRows;
  const int64_t grow = row0 + row;
  float p0 = 0.f, p1 = 0.f, p2 = 0.f;
  if (grow < M) {
    if (grid.enabled) {
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
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    uint32_t vh[4], vl[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
      split_bf16_pair(x3_enc_value(p0, p1, p2, part * 16 + h * 8 + 2 * i), x3_enc_value(p0, p1, p2, part * 16 + h * 8 + 2 * i + 1),
                      vh[i], vl[i]);
    keep[2 * h] = make_uint4(vh[0], vh[1], vh[2], vh[3]);
    keep[2 * h + 1] = make_uint4(vl[0], vl[1], vl[2], vl[3]);
    *reinterpret_cast<uint4*>(ph + row * kLDB + part * 16 + h * 8) = keep[2 * h];
    *reinterpret_cast<uint4*>(pl + row * kLDB + part * 16 + h * 8) = keep[2 * h + 1];
  }
}
__device__ __forceinline__ void posenc_restore_x3(__bf16* __restrict__ ph, __bf16* __restrict__ pl, int tid, const uint4 (&keep)[4]) {
  const int row = tid % kXRows, part = tid / kXRows;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    *reinterpret_cast<uint4*>(ph + row * kLDB + part * 16 + h * 8) = keep[2 * h];
    *reinterpret_cast<uint4*>(pl + row * kLDB + part * 16 + h * 8) = keep[2 * h + 1];
  }
}

// ------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------
template <int NHB, bool RGB>
__global__ __launch_bounds__(kXThreads, kXWgPerCu * kXWaves / 4) void mlp_fwd_x3_kernel(
    const float* __restrict__ pk, const float* __restrict__ pts, X3Grid grid, int64_t M, int deg,
    float* __restrict__ raw_rgb, float* __restrict__ raw_sigma) {
  __shared__ __attribute__((aligned(16))) __bf16 plane_h[kXRows * kLDB];
  __shared__ __attribute__((aligned(16))) __bf16 plane_l[kXRows * kLDB];
  __shared__ __attribute__((aligned(16))) float s_bias[kDepth * kW];      // trunk biases, staged once per workgroup
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int C = rgb_channels(deg);
  const float* __restrict__ bias = pk + fwd_bias_off(deg);
  // this lane's operand position: sample row (lane & 31) of a row block, k offset 8 (lane >> 5) of a k-group
  const __bf16* xh = plane_h + (lane & 31) * kLDB + (lane >> 5) * 8;
  const __bf16* xl = plane_l + (lane & 31) * kLDB + (lane >> 5) * 8;
  const int64_t ntiles = (M + kXRows - 1) / kXRows;
  for (int i = tid; i < kDepth * kW; i += kXThreads) s_bias[i] = bias[i];
  // this wave's weight stream over the trunk: blocks (kg, cb = wave*kXCB + c, part), 2 x 64 f32x4 per (kg, cb);
  // 4 + 16*4 + 20 + 16*2 = 120 k-groups from layer 0 to layer 7
  const X3Image wimg = make_x3image(pk, fwd_image_floats(deg), lane);
  const int wu0 = (wave * kXCB) * 128;              // f32x4 index of this wave's first column block in a k-group
  constexpr int kTrunkKg = 120;

  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t row0 = tile * kXRows;
    __syncthreads();   // previous tile's head GEMM has consumed the planes
    uint4 enc_keep[4];
    posenc_tile_x3(plane_h, plane_l, pts, grid, row0, M, tid, enc_keep);
    __syncthreads();

    f32x16 acc[kXRB][kXCB];
    X3Frag w[4][kXCB];
    int kg0 = 0;          // position of the running layer in the wave's weight stream
    for (int l = 0; l < kDepth; ++l) {
      // the accumulators start from the bias (register quad q of a lane holds features n0 .. n0 + 3 of one sample, whatever the
      // row block): the epilogue saves an add per element on the lanes the MFMAs run on
#pragma unroll
      for (int c = 0; c < kXCB; ++c)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n0 = (wave * kXCB + c) * 32 + 8 * q + 4 * (lane >> 5);
          const f32x4 b4 = *reinterpret_cast<const f32x4*>(s_bias + l * kW + n0);
#pragma unroll
          for (int r = 0; r < kXRB; ++r)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[r][c][4 * q + t] = b4[t];
        }
      const int nkg = l == 0 ? 4 : 16;
      gemm_x3<kXRB, kXCB>(xh, xl, wimg, wu0 + kg0 * (8 * 128), nkg, kTrunkKg - kg0, l > 0, 8 * 128, w, acc);
      kg0 += nkg;
      if (l == 5) {
        // skip connection (model_utils.py:70-71): the 64 encoded columns are a second K segment, put back from registers
        __syncthreads();
        posenc_restore_x3(plane_h, plane_l, tid, enc_keep);
        __syncthreads();
        gemm_x3<kXRB, kXCB>(xh, xl, wimg, wu0 + kg0 * (8 * 128), 4, kTrunkKg - kg0, true, 8 * 128, w, acc);
        kg0 += 4;
      }
      __syncthreads();  // every wave has consumed the input planes
      // epilogue: lane holds features n0 .. n0+3 of sample m per register quad
#pragma unroll
      for (int r = 0; r < kXRB; ++r) {
        const int m = r * 32 + (lane & 31);
#pragma unroll
        for (int c = 0; c < kXCB; ++c)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int n0 = (wave * kXCB + c) * 32 + 8 * q + 4 * (lane >> 5);
            uint32_t h01, l01, h23, l23;
            split_bf16_pair(fmaxf(acc[r][c][4 * q], 0.f), fmaxf(acc[r][c][4 * q + 1], 0.f), h01, l01);
            split_bf16_pair(fmaxf(acc[r][c][4 * q + 2], 0.f), fmaxf(acc[r][c][4 * q + 3], 0.f), h23, l23);
            *reinterpret_cast<uint2*>(plane_h + m * kLDB + n0) = make_uint2(h01, h23);
            *reinterpret_cast<uint2*>(plane_l + m * kLDB + n0) = make_uint2(l01, l23);
          }
      }
      __syncthreads();
    }

    // heads (model_utils.py:72-74, :91-93): wave w owns row block w % kXRB and head blocks w / kXRB, + CSTEP, ...;
    // sigma only (RGB = false): the block that holds column C
    {
      constexpr int CSTEP = kXWaves / kXRB;               // waves per row block (2)
      constexpr int HMAX = RGB ? (NHB + CSTEP - 1) / CSTEP : 1;
      const int rb = wave % kXRB, cb0 = wave / kXRB;
      const float* hb = bias + 8 * kW;
#pragma unroll 1
      for (int i = 0; i < HMAX; ++i) {
        const int cb = RGB ? cb0 + i * CSTEP : (cb0 == 0 ? NHB - 1 : NHB);
        if (cb >= NHB) continue;                           // wave-uniform
        f32x16 hacc[1][1];
        X3Frag hw[4][1];
#pragma unroll
        for (int j = 0; j < 16; ++j) hacc[0][0][j] = 0.f;
        const int wuh = (int)(fwd_layer_off(8) / 4) + cb * 128;
        gemm_x3<1, 1>(xh + rb * 32 * kLDB, xl + rb * 32 * kLDB, wimg, wuh, 16, 16, false, NHB * 128, hw, hacc);
        const int64_t grow = row0 + rb * 32 + (lane & 31);
        if (grow < M) {
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              const int col = cb * 32 + 8 * q + 4 * (lane >> 5) + t;
              const float v = hacc[0][0][4 * q + t] + hb[col];
              if (col < C) { if (RGB) raw_rgb[grow * C + col] = v; }
              else if (col == C) raw_sigma[grow] = v;
            }
        }
      }
    }
  }
}

template <int NHB>
static int launch_x3_nhb(const PxoCfg* cfg, const float* pk, const float* pts, const X3Grid& grid, int64_t M,
                         float* raw_rgb, float* raw_sigma, hipStream_t s) {
  KernelTimer timer(PXO_PROF_MLP_FWD, M, s);
  const int64_t tiles = (M + kXRows - 1) / kXRows, cap = (int64_t)kXWgPerCu * num_cus();
  dim3 grid_dim((unsigned)(tiles < cap ? tiles : cap)), block(kXThreads);
  if (raw_rgb)
    hipLaunchKernelGGL((mlp_fwd_x3_kernel<NHB, true>), grid_dim, block, 0, s, pk, pts, grid, M, cfg->sh_deg, raw_rgb, raw_sigma);
  else
    hipLaunchKernelGGL((mlp_fwd_x3_kernel<NHB, false>), grid_dim, block, 0, s, pk, pts, grid, M, cfg->sh_deg, raw_rgb, raw_sigma);
  return check_launch("mlp_fwd(bf16x3)");
}

int launch_mlp_fwd_x3(const PxoCfg* cfg, const float* packed_fwd, const float* pts, int reso, int x0,
                      const float* off, const float* scale, int64_t M, float* raw_rgb, float* raw_sigma, hipStream_t s) {
  if (M == 0) return PXO_OK;
  X3Grid g;
  g.enabled = pts == nullptr; g.reso = reso > 0 ? reso : 1; g.x0 = x0;
  for (int i = 0; i < 3; ++i) { g.off[i] = off ? off[i] : 0.f; g.scale[i] = scale ? scale[i] : 1.f; }
  switch (head_blocks(cfg->sh_deg)) {
    case 1: return launch_x3_nhb<1>(cfg, packed_fwd, pts, g, M, raw_rgb, raw_sigma, s);
    case 2: return launch_x3_nhb<2>(cfg, packed_fwd, pts, g, M, raw_rgb, raw_sigma, s);
    default: return launch_x3_nhb<3>(cfg, packed_fwd, pts, g, M, raw_rgb, raw_sigma, s);
  }
}

}  // namespace pxo
