// Fused positional-encoding + NeRF-SH MLP forward / backward(data) for gfx950 (MI355X).
//
// Replaces MLP.__call__ of nerf_sh/nerf/model_utils.py:43-94 (torch twin
// octree/nerf/model_utils.py:87-158) and posenc (:145-173) for the use_viewdirs=false /
// SH configuration (nerf_sh/config/blender.yaml, tt.yaml), plus its reverse-mode data path.
//
// Design (exact f32, v_mfma_f32_32x32x2_f32):
//  * one persistent 8-wave workgroup per CU (two waves per SIMD) walks 128-row tiles (pxo_common.h TileSched:
//    whole rounds of full tiles, then one round of 64-row half tiles for a ragged remainder); the 128x256
//    activation tile lives in LDS (row stride 260 floats: conflict-free ds_read_b128 A-fragments) for all 8
//    layers and is updated in place;
//  * every wave owns all 128 rows x 32 columns of a layer (4 accumulator tiles): one weight-fragment load
//    feeds 16 MFMAs.  Weights are pre-packed in MFMA fragment order (pxo_common.h packed_index) so the B
//    operand is one coalesced 16 B/lane load straight from L2 into registers -- the waves' column slices are
//    disjoint, so weights need no LDS staging at all; B is fetched three k-groups ahead through four rotating
//    register sets, A (LDS) one group ahead;
//  * the K order inside a dot product is permuted (lane half h, sub-step j -> k = 8g+4h+j) so
//    that one ds_read_b128 / one 16 B global load feeds four consecutive MFMAs;
//  * post-ReLU activations are streamed to HBM once (for the weight-gradient GEMMs) as whole
//    1 KiB rows copied out of the LDS tile, together with a 1-bit relu mask in fragment order, so
//    the backward-data kernel never re-reads them; bias gradients leave as one [9][256] partial per tile slot
//    (schedule-independent bits; summed in a fixed order by reduce_jobs_kernel).
#include "pxo_common.h"

namespace pxo {

// ------------------------------------------------------------------------------------------
// weight packing
// ------------------------------------------------------------------------------------------
// W_l[k_in][n_out] of the reference layout with zero padding; l == 8 denotes the fused head
// (cols [0,C) = Dense_9, col C = Dense_8).
__device__ __forceinline__ float src_weight(const float* __restrict__ p, int deg, int l, int k, int n) {
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

__global__ void pack_fwd_kernel(const float* __restrict__ p, int deg, float* __restrict__ out) {
  const int nhb = head_blocks(deg);
  const int64_t total = fwd_image_floats(deg);
  const int64_t bias_off = fwd_bias_off(deg);
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    float v;
    if (idx >= bias_off) {
      int b = (int)(idx - bias_off);
      if (b < 8 * kW) {
        v = p[leaf_bias_off(b / kW, deg) + (b % kW)];
      } else {
        int n = b - 8 * kW;
        const int C = rgb_channels(deg);
        v = n < C ? p[leaf_bias_off(9, deg) + n] : (n == C ? p[leaf_bias_off(8, deg)] : 0.f);
      }
    } else {
      int l = 0;
      while (l < 8 && idx >= fwd_layer_off(l + 1)) ++l;
      int64_t loc = idx - fwd_layer_off(l);
      const int ncb = l < 8 ? 8 : nhb;
      int j = (int)(loc & 3), lane = (int)((loc >> 2) & 63);
      int64_t cg = loc >> 8;
      int c = (int)(cg % ncb), g = (int)(cg / ncb);
      int k = 8 * g + 4 * (lane >> 5) + j, n = 32 * c + (lane & 31);
      v = src_weight(p, deg, l, k, n);
    }
    out[idx] = v;
  }
}

__global__ void pack_bwd_kernel(const float* __restrict__ p, int deg, float* __restrict__ out) {
  const int nhb = head_blocks(deg);
  const int64_t total = bwd_image_floats(deg);
  const int64_t head_sz = (int64_t)nhb * 32 * 256;
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    int l;
    int64_t loc;
    if (idx < head_sz) { l = 8; loc = idx; }
    else { int q = (int)((idx - head_sz) / (256 * 256)); l = 7 - q; loc = idx - head_sz - (int64_t)q * 256 * 256; }
    int j = (int)(loc & 3), lane = (int)((loc >> 2) & 63);
    int64_t cg = loc >> 8;
    int c = (int)(cg % 8), g = (int)(cg / 8);
    int k = 8 * g + 4 * (lane >> 5) + j, n = 32 * c + (lane & 31);
    // B^T: contraction index k = forward output column, n = forward input feature (< 256)
    out[idx] = src_weight(p, deg, l, n, k);
  }
}

int launch_pack(const PxoCfg* cfg, const float* mlp_params, float* fwd, float* bwd, hipStream_t s) {
  if (cfg->mlp_precision == PXO_MLP_BF16X6) return launch_pack_x6(cfg, mlp_params, fwd, bwd, s);
  if (cfg->mlp_precision == PXO_MLP_BF16X3) {
    if (bwd) { set_error("pack_weights: mlp_precision bf16x3 is forward-only (packed_bwd must be NULL)"); return PXO_ERR_UNSUPPORTED; }
    return launch_pack_x3(cfg, mlp_params, fwd, s);
  }
  hipLaunchKernelGGL(pack_fwd_kernel, dim3(512), dim3(256), 0, s, mlp_params, cfg->sh_deg, fwd);
  if (bwd) hipLaunchKernelGGL(pack_bwd_kernel, dim3(512), dim3(256), 0, s, mlp_params, cfg->sh_deg, bwd);
  return check_launch("pack_weights");
}

// ------------------------------------------------------------------------------------------
// positional encoding
// ------------------------------------------------------------------------------------------
// column `col` of posenc(p, 0, 10) padded to 64: [p | sin(p*2^l) | sin(p*2^l + pi/2) | 0]
// (nerf_sh/nerf/model_utils.py:160-173, default order: xb index = l*3 + axis).
__device__ __forceinline__ float enc_value(float p0, float p1, float p2, int col) {
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

__global__ void posenc_kernel(const float* __restrict__ x, int64_t N, float* __restrict__ enc) {
  int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (idx >= N * kEnc) return;
  int64_t r = idx / kEnc;
  int col = (int)(idx - r * kEnc);
  enc[idx] = enc_value(x[r * 3], x[r * 3 + 1], x[r * 3 + 2], col);
}

int launch_posenc(const float* x, int64_t N, float* enc, hipStream_t s) {
  if (N == 0) return PXO_OK;
  int64_t total = N * kEnc;
  hipLaunchKernelGGL(posenc_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, N, enc);
  return check_launch("posenc");
}

// dense-grid point source for octree.extraction step1 / auto_scale
// (octree/extraction.py:250-262, :290-303): n -> (ix, iy, iz), x slowest.
struct GridSpec {
  int enabled;
  int reso;
  int x0;
  float off[3];
  float scale[3];
};

__device__ __forceinline__ void grid_point(const GridSpec& g, int64_t n, float& px, float& py, float& pz) {
  const int r = g.reso;
  int iz = (int)(n % r);
  int64_t t = n / r;
  int iy = (int)(t % r);
  int ix = (int)(t / r) + g.x0;
  px = ((((float)ix + 0.5f) / (float)r) - g.off[0]) / g.scale[0];
  py = ((((float)iy + 0.5f) / (float)r) - g.off[1]) / g.scale[1];
  pz = ((((float)iz + 0.5f) / (float)r) - g.off[2]) / g.scale[2];
}

// writes posenc of the tile's 32*RBN points into lds[:, 0:64]; GRID = false (the training kernels: points always come from
// memory) keeps the 12 GridSpec scalars out of the kernel's SGPR budget
// KEEP: also returns this thread's kColsPer values (the forward-only kernels put them back for the skip layer with
// posenc_restore instead of evaluating 16 sinf per point a second time)
template <int RBN> struct EncGeom {
  static constexpr int kRows = 32 * RBN;
  static constexpr int kParts = kMlpThreads / kRows;     // 4 (full tile) / 8 (half tile)
  static constexpr int kColsPer = kEncPad / kParts;      // 16 / 8
};
template <int RBN, bool GRID = true, bool KEEP = false>
__device__ __forceinline__ void posenc_tile(float* __restrict__ lds, const float* __restrict__ pts,
                                            const GridSpec& grid, int64_t row0, int64_t M, int tid,
                                            float (*keep)[EncGeom<RBN>::kColsPer] = nullptr) {
  constexpr int kRows = EncGeom<RBN>::kRows, kColsPer = EncGeom<RBN>::kColsPer;
  const int row = tid % kRows, part = tid / kRows;
  const int64_t grow = row0 + row;
  float p0 = 0.f, p1 = 0.f, p2 = 0.f;
  if (grow < M) {
    if (GRID && grid.enabled) grid_point(grid, grow, p0, p1, p2);
    else { p0 = pts[grow * 3]; p1 = pts[grow * 3 + 1]; p2 = pts[grow * 3 + 2]; }
  }
#pragma unroll 4
  for (int i = 0; i < kColsPer; ++i) {
    const int col = part * kColsPer + i;
    const float e = enc_value(p0, p1, p2, col);
    lds[row * kLDA + col] = e;
    if (KEEP) (*keep)[i] = e;
  }
}
template <int RBN>
__device__ __forceinline__ void posenc_restore(float* __restrict__ lds, int tid, const float (&keep)[EncGeom<RBN>::kColsPer]) {
  constexpr int kRows = EncGeom<RBN>::kRows, kColsPer = EncGeom<RBN>::kColsPer;
  const int row = tid % kRows, part = tid / kRows;
#pragma unroll
  for (int i = 0; i < kColsPer; ++i) lds[row * kLDA + part * kColsPer + i] = keep[i];
}

// ------------------------------------------------------------------------------------------
// the (RBN*32)-row x (CBN*32)-col wave GEMM: A from LDS (ds_read_b128), B from the packed image.
// kgroups must be even; operands are ping-ponged between two register sets (no copies).
// ------------------------------------------------------------------------------------------
template <int RBN, int CBN>
__device__ __forceinline__ void mfma_group(const f32x4 (&a)[RBN], const f32x4 (&b)[CBN], f32x16 (&acc)[RBN][CBN]) {
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < RBN; ++r)
#pragma unroll
      for (int c = 0; c < CBN; ++c)
        acc[r][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r][j], b[c][j], acc[r][c], 0, 0, 0);
}

// Operand addressing of the k-loop without vector ALU work (measured on the skewed-schedule experiment of round 3:
// a VALU instruction issued beside a back-to-back stream of v_mfma_f32_32x32x2_f32 waits for the running MFMA and
// costs the stream ~10 cycles; the loop had 3.5 of them per 16 MFMAs and ran at 95.8 %):
//  * A: ds_read_b128 takes a 16-bit immediate; row blocks 2-3 sit 66,560 B further on, so they get their own offset
//    register (ARows::hi) instead of an add per access;
//  * B: buffer_load_dwordx4 with the weight image as the buffer: the layer / k-group / column-block part of the address
//    is a wave-uniform byte offset (one SGPR, scalar arithmetic), the lane part one 32-bit register that never changes.
struct ARows {
  const float* base;   // the LDS tile
  uint32_t lo, hi;     // float offsets of this lane's fragment row in row blocks 0-1 (+33,280 B immediate) / 2-3
};
__device__ __forceinline__ ARows make_arows(const float* lds, int lane) {
  ARows a{lds, (uint32_t)((lane & 31) * kLDA + (lane >> 5) * 4), 0u};
  a.hi = a.lo + 2 * 32 * kLDA;
  asm volatile("" : "+v"(a.hi));     // an independent induction variable, not "lo + constant" re-derived at every use
  return a;
}
template <int RBN>
__device__ __forceinline__ void load_a(const ARows& ar, int g, f32x4 (&a)[RBN]) {
#pragma unroll
  for (int r = 0; r < RBN; ++r)
    a[r] = *reinterpret_cast<const f32x4*>(ar.base + (r < 2 ? ar.lo : ar.hi) + (r & 1) * 32 * kLDA + g * 8);
}
// the packed weight image as a raw buffer: base + wave-uniform byte offset (SGPR) + lane * 16 (VGPR)
struct WImage {
  __amdgpu_buffer_rsrc_t rsrc;
  uint32_t voff;       // lane * 16
};
// HARDWARE ASSUMPTION shared by every raw-buffer access in this file and in wgrad_kernels.hip: a gfx9 raw buffer
// (stride 0, no swizzle; word 3 = 0x00020000) is range-checked on the SUM voffset + soffset + immediate offset against
// num_records (bytes): a load past the end returns 0, a store past the end is dropped.  Ragged tiles rely on it --
// num_records is set to the tile's valid bytes and the per-row part of the address travels in soffset / the immediate
// (the compiler may move a constant between the two; both are inside the checked sum) -- so rows >= M of the last tile
// are neither read from nor written into memory owned by another tile or layer.  Kept honest by the ragged-M parity tests
// (tests/test_gpu_parity.py: test_mlp_fwd_saved_tensors M = 657 / 300, test_mlp_backward M = 424 / 200, which compare
// acts, dz and every weight gradient with the oracle) -- a toolchain or architecture that checks differently fails them.
__device__ __forceinline__ WImage make_wimage(const float* image, int64_t floats, int lane) {
  return WImage{__builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(image), 0, (int)(floats * 4), 0x00020000),
                (uint32_t)lane * 16u};
}
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// wu: wave-uniform f32x4 index of this wave's first column block of the packed matrix inside the image
template <int CBN>
__device__ __forceinline__ void load_b(const WImage& w, int wu, int g, int kg_stride, f32x4 (&b)[CBN]) {
#pragma unroll
  for (int c = 0; c < CBN; ++c)
    b[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w.rsrc, w.voff, (wu + g * kg_stride + c * 64) * 16, 0));
}

// B (weights, L2 latency) is fetched kBDist k-groups ahead into four rotating register sets (the caller's, primed by
// gemm_prefetch_b), A (LDS) one group ahead into two; kgroups must be a multiple of 4.  No register copies.  The
// sched_barriers pin "issue next loads, then 16 MFMAs": without them hipcc (at the VGPR cap) sinks
// each load to just before its use and exposes the LDS/L2 latency on every k-group.
constexpr int kBDist = 3;   // k-groups of look-ahead for the weight fragments (2 measured 0.25 % slower; four register sets either way)
#define PXO_PIN() __builtin_amdgcn_sched_barrier(0)
// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt: a wave would wait at every barrier
// for the acknowledgement of the activation-tile stores it has just issued (and for its prefetched weight fragments);
// nothing that crosses waves inside these kernels lives in global memory.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
// the first kBDist weight fragments of a GEMM (issued by the caller ahead of time, e.g. before the previous layer's
// epilogue, so that their L2 latency is not exposed when the loop starts)
template <int CBN>
__device__ __forceinline__ void gemm_prefetch_b(const WImage& w, int wu, int kgroups, int kg_stride,
                                                f32x4 (&b)[4][CBN]) {
  const int last = kgroups - 1;
#pragma unroll
  for (int i = 0; i < kBDist; ++i) load_b<CBN>(w, wu, i < last ? i : last, kg_stride, b[i]);
}

// Copy of this wave's 32 columns of the tile the GEMM is reading (the previous layer's output) to HBM, one 8-row slab per
// k-group during the first 4*RBN k-groups of the loop: the store path (64 B/clk per CU, 2048 cycles for a 128 KB tile) then
// works beside the MFMAs instead of between the epilogue and the next GEMM.  No vector ALU work: LDS reads with immediate
// offsets off two base registers, buffer stores with scalar row offsets.
struct TileCopy {
  __amdgpu_buffer_rsrc_t out;   // the tile's valid rows of the destination array
  uint32_t voff;                // (rsub * 256 + wave * 32 + c4 * 4) * 4
  const float* src;             // LDS: this lane's piece of slab 0
  uint32_t src_hi;              // float offset of slab 8 (its own register: beyond the ds_read immediate range)
  bool on;
};
template <int RBN>
__device__ __forceinline__ TileCopy make_tile_copy(const float* lds, float* dst, int64_t row0, int64_t M, int wave, int lane,
                                                   bool on) {
  const int c4 = lane & 7, rsub = lane >> 3;
  const int64_t rows = M - row0 < 32 * RBN ? M - row0 : 32 * RBN;
  TileCopy t;
  t.out = __builtin_amdgcn_make_buffer_rsrc(dst + row0 * kW, 0, on ? (int)(rows * kW * 4) : 0, 0x00020000);
  t.voff = (uint32_t)(rsub * kW + wave * 32 + c4 * 4) * 4u;
  t.src = lds + rsub * kLDA + wave * 32 + c4 * 4;
  t.src_hi = 8 * 8 * kLDA;
  asm volatile("" : "+v"(t.src_hi));
  t.on = on;
  return t;
}
template <int RBN>
__device__ __forceinline__ void tile_copy_slab(const TileCopy& t, int i) {      // i: compile-time after unrolling
  const f32x4 v = i < 8 ? *reinterpret_cast<const f32x4*>(t.src + i * 8 * kLDA)
                        : *reinterpret_cast<const f32x4*>(t.src + t.src_hi + (i - 8) * 8 * kLDA);
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), t.out, t.voff, i * 8 * kW * 4, 0);
}

template <int RBN, int CBN, bool COPY = false>
__device__ __forceinline__ void gemm_lds_packed(const ARows& arow, const WImage& w, int wu,
                                                int kgroups, int kg_stride, f32x16 (&acc)[RBN][CBN],
                                                f32x4 (&b)[4][CBN], const TileCopy* tc = nullptr) {
  constexpr int D = kBDist;
  static_assert(D == 2 || D == 3, "kBDist");
  f32x4 a0[RBN], a1[RBN];
  const int last = kgroups - 1;
  auto cl = [&](int g) { return g < last ? g : last; };      // harmless re-loads past the end
  load_a<RBN>(arow, 0, a0);
  for (int g = 0; g < kgroups; g += 4) {
    // phase p multiplies k-group g+p out of set p and refills set (p+D)%4 with k-group g+p+D
    load_a<RBN>(arow, g + 1, a1);
    load_b<CBN>(w, wu, cl(g + D), kg_stride, b[D & 3]);
    PXO_PIN();
    mfma_group<RBN, CBN>(a0, b[0], acc);
    PXO_PIN();
    if (COPY && tc->on && g < 4 * RBN) {
      switch (g >> 2) {
        case 0: tile_copy_slab<RBN>(*tc, 0); break;
        case 1: tile_copy_slab<RBN>(*tc, 4 + 0); break;
        case 2: if (RBN > 2) tile_copy_slab<RBN>(*tc, 8 + 0); break;
        default: if (RBN > 2) tile_copy_slab<RBN>(*tc, 12 + 0); break;
      }
      PXO_PIN();
    }
    load_a<RBN>(arow, g + 2, a0);
    load_b<CBN>(w, wu, cl(g + 1 + D), kg_stride, b[(1 + D) & 3]);
    PXO_PIN();
    mfma_group<RBN, CBN>(a1, b[1], acc);
    PXO_PIN();
    if (COPY && tc->on && g < 4 * RBN) {
      switch (g >> 2) {
        case 0: tile_copy_slab<RBN>(*tc, 1); break;
        case 1: tile_copy_slab<RBN>(*tc, 4 + 1); break;
        case 2: if (RBN > 2) tile_copy_slab<RBN>(*tc, 8 + 1); break;
        default: if (RBN > 2) tile_copy_slab<RBN>(*tc, 12 + 1); break;
      }
      PXO_PIN();
    }
    load_a<RBN>(arow, g + 3, a1);
    load_b<CBN>(w, wu, cl(g + 2 + D), kg_stride, b[(2 + D) & 3]);
    PXO_PIN();
    mfma_group<RBN, CBN>(a0, b[2], acc);
    PXO_PIN();
    if (COPY && tc->on && g < 4 * RBN) {
      switch (g >> 2) {
        case 0: tile_copy_slab<RBN>(*tc, 2); break;
        case 1: tile_copy_slab<RBN>(*tc, 4 + 2); break;
        case 2: if (RBN > 2) tile_copy_slab<RBN>(*tc, 8 + 2); break;
        default: if (RBN > 2) tile_copy_slab<RBN>(*tc, 12 + 2); break;
      }
      PXO_PIN();
    }
    load_a<RBN>(arow, cl(g + 4), a0);
    load_b<CBN>(w, wu, cl(g + 3 + D), kg_stride, b[(3 + D) & 3]);
    PXO_PIN();
    mfma_group<RBN, CBN>(a1, b[3], acc);
    PXO_PIN();
    if (COPY && tc->on && g < 4 * RBN) {
      switch (g >> 2) {
        case 0: tile_copy_slab<RBN>(*tc, 3); break;
        case 1: tile_copy_slab<RBN>(*tc, 4 + 3); break;
        case 2: if (RBN > 2) tile_copy_slab<RBN>(*tc, 8 + 3); break;
        default: if (RBN > 2) tile_copy_slab<RBN>(*tc, 12 + 3); break;
      }
      PXO_PIN();
    }
  }
}

// head GEMM: one row block, CBN column blocks `cb_stride` f32x4 apart, 32 k-groups, pipelined like the trunk
template <int CBN>
__device__ __forceinline__ void gemm_head(const float* __restrict__ arow, const f32x4* __restrict__ wp, int cb_stride,
                                          int kg_stride, f32x16 (&acc)[1][CBN]) {
  f32x4 a0[1], a1[1], b0[CBN], b1[CBN], b2[CBN], b3[CBN];
  auto lb = [&](int g, f32x4 (&b)[CBN]) {
#pragma unroll
    for (int c = 0; c < CBN; ++c) b[c] = wp[(int64_t)g * kg_stride + c * cb_stride];
  };
  auto la = [&](int g, f32x4 (&a)[1]) { a[0] = *reinterpret_cast<const f32x4*>(arow + g * 8); };
  lb(0, b0); lb(1, b1);
  la(0, a0);
  for (int g = 0; g < 32; g += 4) {
    la(g + 1, a1); lb(g + 2, b2); PXO_PIN();
    mfma_group<1, CBN>(a0, b0, acc); PXO_PIN();
    la(g + 2, a0); lb(g + 3, b3); PXO_PIN();
    mfma_group<1, CBN>(a1, b1, acc); PXO_PIN();
    la(g + 3, a1); lb(g + 4 < 31 ? g + 4 : 31, b0); PXO_PIN();
    mfma_group<1, CBN>(a0, b2, acc); PXO_PIN();
    la(g + 4 < 31 ? g + 4 : 31, a0); lb(g + 5 < 31 ? g + 5 : 31, b1); PXO_PIN();
    mfma_group<1, CBN>(a1, b3, acc); PXO_PIN();
  }
}

template <int RBN, int CBN>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[RBN][CBN]) {
#pragma unroll
  for (int r = 0; r < RBN; ++r)
#pragma unroll
    for (int c = 0; c < CBN; ++c)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[r][c][i] = 0.f;
}

// accumulator register `reg` of a 32x32 tile holds row (reg&3) + 8*(reg>>2) + 4*(lane>>5)
__device__ __forceinline__ int frag_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

// Copy of ONE wave's 32 columns of the finished (32*RBN) x 256 LDS tile to a row-major [M,256] global array, by the wave
// that has just written them (no cross-wave ordering needed: a wave's LDS operations execute in order): per
// instruction 8 rows x 128 B (whole 128-byte lines).  A workgroup-wide copy of whole 1 KiB rows needs a barrier between
// the epilogue and the copy; without it the copy is issued right behind the wave's LDS writes and the forward / backward
// kernels run 1.0 % / 2.9 % faster (round 3 A/B).  The destination is a buffer bounded to the tile's valid rows: rows
// past M are dropped by the bounds check, the row offset of every store is a scalar -- no predicates, no vector address math.
template <int RBN>
__device__ __forceinline__ void store_wave_cols(const float* __restrict__ lds, float* __restrict__ dst, int64_t row0,
                                                int64_t M, int wave, int lane) {
  const int c4 = lane & 7, rsub = lane >> 3;
  const int64_t rows = M - row0 < 32 * RBN ? M - row0 : 32 * RBN;
  const __amdgpu_buffer_rsrc_t out = __builtin_amdgcn_make_buffer_rsrc(dst + row0 * kW, 0, (int)(rows * kW * 4), 0x00020000);
  const uint32_t voff = (uint32_t)(rsub * kW + wave * 32 + c4 * 4) * 4u;
  const float* __restrict__ src = lds + rsub * kLDA + wave * 32 + c4 * 4;
#pragma unroll
  for (int i0 = 0; i0 < 4 * RBN; i0 += 4) {      // four LDS reads in flight per four stores
    f32x4 v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const f32x4*>(src + (i0 + i) * 8 * kLDA);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v[i]), out, voff, (i0 + i) * 8 * kW * 4, 0);
  }
}

// relu mask, one bit per accumulator element in (row block, column block, register) order, packed MSB-first:
// the forward epilogue shifts the bit "v > 0" in from the carry (mw = 2 mw + bit: compare + add-with-carry), the
// backward epilogue shifts it out again into the carry that selects the gradient (add + select).
__device__ __forceinline__ void mask_push(uint32_t& mw, float v) {
  asm volatile("v_cmp_lt_f32 vcc, 0, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(mw) : "v"(v) : "vcc");
}
__device__ __forceinline__ float mask_pop(uint32_t& mw, float x) {
  float r;
  asm volatile("v_add_co_u32 %0, vcc, %0, %0\n\tv_cndmask_b32 %1, 0, %2, vcc" : "+v"(mw), "=v"(r) : "v"(x) : "vcc");
  return r;
}

constexpr int kRB = kTM / 32;                    // row blocks of a full tile (all owned by every wave)
constexpr int kCB = 8 / kMlpWaves;               // column blocks per wave in a 256-wide layer (1)

// ------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------
// One tile of 32*RBN rows starting at row0 (RBN = 4: full tile, RBN = 2: half-height tail tile, see
// pxo_common.h TileSched) through posenc, the 8 trunk layers and the heads.  RGB = false computes only the
// head block that holds the sigma column (octree/extraction.py:316-317 and the sparsity branch of
// nerf_sh/train.py:80-81 discard raw_rgb).
template <int NHB, bool SAVE, bool RGB, int RBN>
__device__ __forceinline__ void fwd_tile(float* __restrict__ lds, const float* __restrict__ pk,
                                         const float* __restrict__ pts, const GridSpec& grid, int64_t M, int deg,
                                         int64_t row0, int64_t slot, float* __restrict__ raw_rgb,
                                         float* __restrict__ raw_sigma, float* __restrict__ acts,
                                         float* __restrict__ enc_out, uint32_t* __restrict__ mask, int tid, int lane,
                                         int wave) {
  constexpr int kRows = 32 * RBN;
  constexpr int kWordsUsed = RBN * kCB * 16 / 32;     // relu-mask words this tile height fills (of kMaskWords)
  // per-tile opaque copies of the thread ids: whatever is derived from them (store / load addresses of every phase) is
  // computed inside the tile instead of being hoisted out of the persistent loop into registers that stay live through
  // the GEMMs
  asm volatile("" : "+v"(tid));
  lane = tid & 63;
  const int C = rgb_channels(deg);
  const float* __restrict__ bias = pk + fwd_bias_off(deg);
  const float* arow1 = lds + (lane & 31) * kLDA + (lane >> 5) * 4;
  const ARows arow = make_arows(lds, lane);
  const WImage wimg = make_wimage(pk, fwd_image_floats(deg), lane);
  const bool full = row0 + kRows <= M;
  lds_barrier();   // previous tile's head GEMM has consumed the LDS tile
  float enc_keep[EncGeom<RBN>::kColsPer];
  posenc_tile<RBN, !SAVE, !SAVE>(lds, pts, grid, row0, M, tid, &enc_keep);
  lds_barrier();
  if (SAVE) {  // coalesced copy of the encoded tile (layer-0 / layer-5 weight gradients)
#pragma unroll
    for (int i = 0; i < kRows * kEncPad / 4 / kMlpThreads; ++i) {
      const int idx = tid + kMlpThreads * i;
      const int row = idx >> 4, c4 = idx & 15;
      if (full || row0 + row < M)
        *reinterpret_cast<f32x4*>(enc_out + (row0 + row) * kEncPad + c4 * 4) =
            *reinterpret_cast<const f32x4*>(lds + row * kLDA + c4 * 4);
    }
  }

  f32x16 acc[RBN][kCB];
  f32x4 bfrag[4][kCB];
  auto layer_wp = [&](int l) {        // wave-uniform f32x4 index into the image
    return (int)(fwd_layer_off(l) / 4) + (wave * kCB) * 64;
  };
  gemm_prefetch_b<kCB>(wimg, layer_wp(0), 8, 8 * 64, bfrag);
  float bl[kCB];                         // the coming layer's biases (this lane's column), fetched one layer ahead
#pragma unroll
  for (int c = 0; c < kCB; ++c) bl[c] = bias[(wave * kCB + c) * 32 + (lane & 31)];
  for (int l = 0; l < kDepth; ++l) {
    // the accumulators start from the bias (every register of a 32x32 fragment holds this lane's column): the epilogue
    // saves its 64 adds per wave and layer, which run on the same f32 lanes as the MFMAs
#pragma unroll
    for (int r = 0; r < RBN; ++r)
#pragma unroll
      for (int c = 0; c < kCB; ++c)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[r][c][i] = bl[c];
    const int wp = layer_wp(l);
    if (SAVE && l > 0) {
      // the previous layer's activations leave for HBM during the GEMM that reads them (TileCopy)
      int lane_c = lane;
      asm volatile("" : "+v"(lane_c));   // the copy's lane constants are derived here, not hoisted over the posenc phase
      const TileCopy tc = make_tile_copy<RBN>(lds, acts + (int64_t)(l - 1) * M * kW, row0, M, wave, lane_c, true);
      gemm_lds_packed<RBN, kCB, true>(arow, wimg, wp, 32, 8 * 64, acc, bfrag, &tc);
    } else {
      gemm_lds_packed<RBN, kCB>(arow, wimg, wp, l == 0 ? 8 : 32, 8 * 64, acc, bfrag);
    }
    if (l == 5) {
      // skip connection (model_utils.py:70-71): x = concat([h4, inputs]) -> the 64 encoded
      // columns are a second K segment; the encoding is recomputed into the consumed tile.
      gemm_prefetch_b<kCB>(wimg, wp + 32 * 8 * 64, 8, 8 * 64, bfrag);
      lds_barrier();
      if (SAVE) {
        // training: every thread reads back the 16-byte pieces of the encoded tile it stored itself at the start of the
        // tile (L2-resident), ~20 instructions instead of 16 sinf evaluations on the lanes the MFMAs run on
#pragma unroll
        for (int i = 0; i < kRows * kEncPad / 4 / kMlpThreads; ++i) {
          const int idx = tid + kMlpThreads * i;
          const int row = idx >> 4, c4 = idx & 15;
          f32x4 v = {0.f, 0.f, 0.f, 0.f};
          if (full || row0 + row < M) v = *reinterpret_cast<const f32x4*>(enc_out + (row0 + row) * kEncPad + c4 * 4);
          *reinterpret_cast<f32x4*>(lds + row * kLDA + c4 * 4) = v;
        }
      } else {
        posenc_restore<RBN>(lds, tid, enc_keep);
      }
      lds_barrier();
      gemm_lds_packed<RBN, kCB>(arow, wimg, wp + 32 * 8 * 64, 8, 8 * 64, acc, bfrag);
    }
    // the next GEMM's first weight fragments (and biases) travel while this wave is in its epilogue
    if (l + 1 < kDepth) {
      gemm_prefetch_b<kCB>(wimg, layer_wp(l + 1), 32, 8 * 64, bfrag);
#pragma unroll
      for (int c = 0; c < kCB; ++c) bl[c] = bias[(l + 1) * kW + (wave * kCB + c) * 32 + (lane & 31)];
    }
    lds_barrier();  // every wave has consumed the columns this wave is about to rewrite
    // re-derive the lane ids from an opaque copy so that the epilogue / store addresses are
    // computed here instead of being hoisted out of the loops into (scarce) registers
    int tid_e = tid;
    asm volatile("" : "+v"(tid_e));
    const int lane_e = tid_e & 63;
    uint32_t mw[kMaskWords];
#pragma unroll
    for (int w = 0; w < kMaskWords; ++w) mw[w] = 0u;
#pragma unroll
    for (int r = 0; r < RBN; ++r)
#pragma unroll
      for (int c = 0; c < kCB; ++c) {
        const int col = (wave * kCB + c) * 32 + (lane_e & 31);
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int row = r * 32 + frag_row(reg, lane_e);
          const float v = fmaxf(acc[r][c][reg], 0.f);
          lds[row * kLDA + col] = v;
          if (SAVE) mask_push(mw[((r * kCB + c) * 16 + reg) >> 5], v);
        }
      }
    if (SAVE) {
      uint32_t* mp = mask + ((slot * kDepth + l) * kMlpThreads + tid_e) * kMaskWords;
#pragma unroll
      for (int w = 0; w < kWordsUsed; ++w) mp[w] = mw[w];
      // layers 0..6 leave for HBM during the next layer's GEMM (TileCopy above: 4.08 -> 3.99 ms per launch; a first attempt
      // measured 2.3 % SLOWER because the copy's lane constants were hoisted out of the persistent tile loop and pushed the
      // kernel into spills -- the per-tile opaque thread id at the top of this function is what makes it pay); only the last
      // layer has no trunk GEMM behind it: its 32 columns leave right behind this wave's LDS writes (no barrier in between)
      if (l == kDepth - 1) store_wave_cols<RBN>(lds, acts + (int64_t)l * M * kW, row0, M, wave, lane_e);
    }
    lds_barrier();
  }

  // heads: [raw_rgb | raw_sigma] = h7 @ [Dense_9 | Dense_8] + b (model_utils.py:72-74, :91-93);
  // wave w owns row block w % RBN and the column blocks w / RBN, w / RBN + CSTEP, ... -- one 32x32
  // accumulator at a time (two at once, as SH25's three blocks over two waves would need, spill the
  // 256-VGPR budget of this kernel)
  {
    constexpr int CSTEP = kMlpWaves / RBN;               // waves sharing a row block
    constexpr int HMAX = RGB ? (NHB + CSTEP - 1) / CSTEP : 1;
    const int rb = wave % RBN, cb0 = wave / RBN;
    const float* ar = arow1 + rb * 32 * kLDA;
    const float* hb = bias + 8 * kW;
#pragma unroll 1
    for (int i = 0; i < HMAX; ++i) {
      // sigma-only: the last block holds column C (Dense_8); one wave per row block computes it
      const int cb = RGB ? cb0 + i * CSTEP : (cb0 == 0 ? NHB - 1 : NHB);
      if (cb >= NHB) continue;                             // wave-uniform
      f32x16 hacc[1][1];
      zero_acc(hacc);
      const f32x4* wp = reinterpret_cast<const f32x4*>(pk + fwd_layer_off(8)) + cb * 64 + lane;
      gemm_head<1>(ar, wp, 0, NHB * 64, hacc);
      const int col = cb * 32 + (lane & 31);
      const float b = hb[col];
      // one destination per lane (its head column), chosen once: the 16 stores below are then straight-line code (per-
      // element branches made hipcc wait for every store before the next one)
      float* __restrict__ dst = nullptr;
      int64_t stride = 0;
      if (col < C) { if (RGB) { dst = raw_rgb + col; stride = C; } }
      else if (col == C) { dst = raw_sigma; stride = 1; }
      if (dst) {
        const int64_t r0 = row0 + rb * 32 + 4 * (lane >> 5);
        if (full) {
#pragma unroll
          for (int reg = 0; reg < 16; ++reg) dst[(r0 + (reg & 3) + 8 * (reg >> 2)) * stride] = hacc[0][0][reg] + b;
        } else {
#pragma unroll
          for (int reg = 0; reg < 16; ++reg) {
            const int64_t grow = r0 + (reg & 3) + 8 * (reg >> 2);
            if (grow < M) dst[grow * stride] = hacc[0][0][reg] + b;
          }
        }
      }
    }
  }
}

// Tile schedules of the persistent workgroups (both kernels of this file):
//   static  (DYN = false)  workgroup b runs slots b, b + grid, b + 2 grid, ...
//   dynamic (DYN = true)   workgroup b runs slot b first and then TAKES slots from a device counter (zero at launch):
//                          slot = grid + atomicAdd(counter, 1).  The ticket for the NEXT slot is drawn by thread 0 when a
//                          tile starts (the atomic's round trip rides with the tile's first loads) and handed over
//                          through LDS when the tile ends: the schedule costs two LDS barriers per tile.
// Results do not depend on the schedule: a slot's rows, relu-mask words and bias partial are a function of the slot alone.
// Why dynamic: with one 8-wave workgroup per CU at the full register / LDS budget nothing else can be co-resident, so a
// collective's kernel on the high-priority exchange stream (dist.GradReducer) takes whole CUs when it starts at a kernel
// boundary; under the static stride the workgroups that start late on those CUs become the launch's tail, under the counter
// their share is absorbed by all the others (scripts/contention_probe.py).  Skipping mode (bwd) needs it for load balance.
struct TileTicket {
  int* next;                       // LDS word
  unsigned int* counter;           // device word, zero when the launch starts
  // thread 0 draws the ticket when the tile starts and keeps it in a register: the atomic's round trip then overlaps the
  // tile's first loads (its return is waited for with them) instead of holding thread 0's wave -- and with it the whole
  // workgroup at the tile's first barrier -- for ~1 us
  __device__ __forceinline__ int draw(int tid) const {
    return tid == 0 ? (int)gridDim.x + (int)atomicAdd(counter, 1u) : 0;
  }
  __device__ __forceinline__ int64_t take(int tid, int ticket) const {
    if (tid == 0) *next = ticket;
    lds_barrier();                 // the ticket is in LDS (and every wave is through the tile)
    const int v = __builtin_amdgcn_readfirstlane(*next);
    lds_barrier();                 // everyone has read it before thread 0 overwrites it
    return v;
  }
};

template <int NHB, bool SAVE, bool RGB, bool DYN = false>
__global__ __launch_bounds__(kMlpThreads, kMlpWgPerCu * kMlpWaves / 4) void mlp_fwd_kernel(
    const float* __restrict__ pk, const float* __restrict__ pts, GridSpec grid, int64_t M, int deg, TileSched ts,
    float* __restrict__ raw_rgb, float* __restrict__ raw_sigma, float* __restrict__ acts,
    float* __restrict__ enc_out, uint32_t* __restrict__ mask, unsigned int* __restrict__ tile_counter) {
  __shared__ __attribute__((aligned(16))) float lds[kTM * kLDA + (DYN ? 4 : 0)];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (!DYN) {
    for (int64_t tile = blockIdx.x; tile < ts.n_full; tile += gridDim.x)
      fwd_tile<NHB, SAVE, RGB, kRB>(lds, pk, pts, grid, M, deg, tile * kTM, tile, raw_rgb, raw_sigma, acts, enc_out,
                                    mask, tid, lane, wave);
    for (int64_t h = blockIdx.x; h < ts.n_half; h += gridDim.x)
      fwd_tile<NHB, SAVE, RGB, kRB / 2>(lds, pk, pts, grid, M, deg, ts.half_row0 + h * (kTM / 2), ts.n_full + h,
                                        raw_rgb, raw_sigma, acts, enc_out, mask, tid, lane, wave);
  } else {
    const TileTicket tk{reinterpret_cast<int*>(lds + kTM * kLDA), tile_counter};
    const int64_t n_slots = ts.n_full + ts.n_half;
    for (int64_t slot = blockIdx.x; slot < n_slots;) {
      const int ticket = tk.draw(tid);
      if (slot < ts.n_full)
        fwd_tile<NHB, SAVE, RGB, kRB>(lds, pk, pts, grid, M, deg, slot * kTM, slot, raw_rgb, raw_sigma, acts, enc_out,
                                      mask, tid, lane, wave);
      else
        fwd_tile<NHB, SAVE, RGB, kRB / 2>(lds, pk, pts, grid, M, deg, ts.half_row0 + (slot - ts.n_full) * (kTM / 2), slot,
                                          raw_rgb, raw_sigma, acts, enc_out, mask, tid, lane, wave);
      slot = tk.take(tid, ticket);
    }
  }
}

static unsigned mlp_grid(int64_t M) {
  const int64_t tiles = num_tiles(M), cap = kMlpWgPerCu * (int64_t)num_cus();
  return (unsigned)(tiles < cap ? tiles : cap);
}

template <int NHB>
static int launch_fwd_nhb(const PxoCfg* cfg, const float* pk, const float* pts, const GridSpec& grid,
                          int64_t M, float* raw_rgb, float* raw_sigma, float* acts, float* enc,
                          uint32_t* mask, unsigned int* tile_counter, hipStream_t s) {
  KernelTimer timer(PXO_PROF_MLP_FWD, M, s);
  dim3 grid_dim(mlp_grid(M)), block(kMlpThreads);
  const TileSched ts = tile_sched(M, grid_dim.x);
  // the training instantiations (saved tensors) take their tiles from `tile_counter` when the caller provides one
  // (pre-zeroed, see launch_uniform_jobs); the forward-only ones keep the static stride
  if (acts && raw_rgb && tile_counter)
    hipLaunchKernelGGL((mlp_fwd_kernel<NHB, true, true, true>), grid_dim, block, 0, s, pk, pts, grid, M, cfg->sh_deg, ts,
                       raw_rgb, raw_sigma, acts, enc, mask, tile_counter);
  else if (acts && raw_rgb)
    hipLaunchKernelGGL((mlp_fwd_kernel<NHB, true, true>), grid_dim, block, 0, s, pk, pts, grid, M, cfg->sh_deg, ts,
                       raw_rgb, raw_sigma, acts, enc, mask, tile_counter);
  else if (acts)
    hipLaunchKernelGGL((mlp_fwd_kernel<NHB, true, false>), grid_dim, block, 0, s, pk, pts, grid, M, cfg->sh_deg, ts,
                       raw_rgb, raw_sigma, acts, enc, mask, tile_counter);
  else if (raw_rgb)
    hipLaunchKernelGGL((mlp_fwd_kernel<NHB, false, true>), grid_dim, block, 0, s, pk, pts, grid, M, cfg->sh_deg, ts,
                       raw_rgb, raw_sigma, acts, enc, mask, tile_counter);
  else
    hipLaunchKernelGGL((mlp_fwd_kernel<NHB, false, false>), grid_dim, block, 0, s, pk, pts, grid, M, cfg->sh_deg, ts,
                       raw_rgb, raw_sigma, acts, enc, mask, tile_counter);
  return check_launch("mlp_fwd");
}

static int launch_fwd_any(const PxoCfg* cfg, const float* pk, const float* pts, const GridSpec& grid,
                          int64_t M, float* raw_rgb, float* raw_sigma, float* acts, float* enc,
                          uint32_t* mask, unsigned int* tile_counter, hipStream_t s) {
  if (M == 0) return PXO_OK;
  switch (head_blocks(cfg->sh_deg)) {
    case 1: return launch_fwd_nhb<1>(cfg, pk, pts, grid, M, raw_rgb, raw_sigma, acts, enc, mask, tile_counter, s);
    case 2: return launch_fwd_nhb<2>(cfg, pk, pts, grid, M, raw_rgb, raw_sigma, acts, enc, mask, tile_counter, s);
    default: return launch_fwd_nhb<3>(cfg, pk, pts, grid, M, raw_rgb, raw_sigma, acts, enc, mask, tile_counter, s);
  }
}

int launch_mlp_fwd(const PxoCfg* cfg, const float* packed_fwd, const float* pts, int64_t M,
                   float* raw_rgb, float* raw_sigma, float* acts, float* enc, uint32_t* mask,
                   hipStream_t s, unsigned int* tile_counter) {
  if (cfg->mlp_precision == PXO_MLP_BF16X6)
    return launch_mlp_fwd_x6(cfg, packed_fwd, pts, 0, 0, nullptr, nullptr, M, raw_rgb, raw_sigma, acts, enc, mask, tile_counter, s);
  if (cfg->mlp_precision == PXO_MLP_BF16X3) {
    if (acts || enc || mask) { set_error("mlp_fwd: mlp_precision bf16x3 is inference-only (no saved tensors)"); return PXO_ERR_UNSUPPORTED; }
    return launch_mlp_fwd_x3(cfg, packed_fwd, pts, 0, 0, nullptr, nullptr, M, raw_rgb, raw_sigma, s);
  }
  GridSpec g;
  g.enabled = 0; g.reso = 1; g.x0 = 0;
  for (int i = 0; i < 3; ++i) { g.off[i] = 0.f; g.scale[i] = 1.f; }
  return launch_fwd_any(cfg, packed_fwd, pts, g, M, raw_rgb, raw_sigma, acts, enc, mask, tile_counter, s);
}

int launch_mlp_fwd_grid(const PxoCfg* cfg, const float* packed_fwd, int reso, int x0, int x1,
                        const float* off, const float* scale, float* sigma_out, hipStream_t s) {
  if (cfg->mlp_precision == PXO_MLP_BF16X6)
    return launch_mlp_fwd_x6(cfg, packed_fwd, nullptr, reso, x0, off, scale, (int64_t)(x1 - x0) * reso * reso, nullptr, sigma_out,
                             nullptr, nullptr, nullptr, nullptr, s);
  if (cfg->mlp_precision == PXO_MLP_BF16X3)
    return launch_mlp_fwd_x3(cfg, packed_fwd, nullptr, reso, x0, off, scale, (int64_t)(x1 - x0) * reso * reso, nullptr,
                             sigma_out, s);
  GridSpec g;
  g.enabled = 1; g.reso = reso; g.x0 = x0;
  for (int i = 0; i < 3; ++i) { g.off[i] = off[i]; g.scale[i] = scale[i]; }
  const int64_t M = (int64_t)(x1 - x0) * reso * reso;
  return launch_fwd_any(cfg, packed_fwd, nullptr, g, M, nullptr, sigma_out, nullptr, nullptr, nullptr, nullptr, s);
}

// ------------------------------------------------------------------------------------------
// backward (data): d_raw -> dz_7 .. dz_0
// ------------------------------------------------------------------------------------------
template <int NHB, int RBN, bool SKIP>
__device__ __forceinline__ int bwd_tile(float* __restrict__ lds, float* __restrict__ db_slot, uint8_t* __restrict__ tile_live,
                                         const float* __restrict__ pkb, const float* __restrict__ d_raw_rgb,
                                         const float* __restrict__ d_raw_sigma, const uint32_t* __restrict__ mask,
                                         int64_t M, int deg, int64_t row0, int64_t slot, float* __restrict__ dz,
                                         int* __restrict__ nz, uint8_t* __restrict__ chunk_live,
                                         int tid, int lane, int wave, int mask_word0 = 0) {
  // returns 0 when the tile is done (or skipped); in skipping mode a FULL tile whose live rows all sit in one 64-row half
  // returns 1 (lower half) / 2 (upper half) without computing: the caller runs that half as a half-height tile (mask_word0
  // = which of the slot's two mask words holds the half's bits).  The dead half adds exact zeros to the tile's bias column
  // sums, in front of or behind the live half's terms, so the slot's partial keeps its bits.
  constexpr int NH = 32 * NHB;
  constexpr int kRows = 32 * RBN;
  constexpr int kWordsUsed = RBN * kCB * 16 / 32;
  constexpr int kChunks = kRows / kLiveRows;
  const int C = rgb_channels(deg);
  const ARows arow = make_arows(lds, lane);
  const WImage wimg = make_wimage(pkb, bwd_image_floats(deg), lane);
  lds_barrier();   // previous tile's stores out of LDS are done
  if (SKIP && tid < kChunks) nz[tid] = 0;
  if (SKIP) lds_barrier();
  // d_raw tile -> lds[:, 0:NH] with the head's column order (d_raw_rgb == NULL: sigma-only rows)
  for (int idx = tid; idx < kRows * NH; idx += kMlpThreads) {
    const int row = idx / NH, col = idx - row * NH;
    const int64_t grow = row0 + row;
    float v = 0.f;
    if (grow < M) {
      if (col < C) { if (d_raw_rgb) v = d_raw_rgb[grow * C + col]; }
      else if (col == C) v = d_raw_sigma[grow];
    }
    lds[row * kLDA + col] = v;
    if (SKIP && v != 0.f) nz[row / kLiveRows] = 1;                // same value from every writer
  }
  lds_barrier();
  if (SKIP) {
    // Rows whose upstream gradient (d_raw_rgb, d_raw_sigma) is exactly zero -- samples in empty space (relu(sigma) = 0
    // and weight 0), samples behind an opaque surface, every sample of a background ray -- have dz_l = 0 in every layer
    // and add exactly 0 to every weight and bias gradient.  Per 16-row chunk a flag says whether any row is live: the
    // weight-gradient kernels skip dead chunks (they then never read this tile's dz), and a tile without a live chunk is
    // skipped here altogether.  Results are bit-identical to the dense pass (sums lose only exact zeros).
    int any = 0;
#pragma unroll
    for (int c = 0; c < kChunks; ++c) any |= nz[c];
    if (tid < kChunks) chunk_live[row0 / kLiveRows + tid] = (uint8_t)nz[tid];
    if (tid == 0) *tile_live = (uint8_t)(any != 0);       // the reduction of the bias partials leaves dead slots out
    if (!any) return 0;
    if (RBN == kRB) {
      int lo = 0, hi = 0;
#pragma unroll
      for (int c = 0; c < kChunks / 2; ++c) { lo |= nz[c]; hi |= nz[kChunks / 2 + c]; }
      if (!hi) return 1;
      if (!lo) return 2;
    }
  } else if (tid == 0) {
    *tile_live = 1;
  }
  if (tid < NH) {  // head bias gradient: this tile's column sums (one [9][256] partial per tile slot)
    float sum = 0.f;
#pragma unroll 8
    for (int row = 0; row < kRows; ++row) sum += lds[row * kLDA + tid];
    db_slot[8 * kW + tid] = sum;
  }
  if (NH < kW && tid >= NH && tid < kW) db_slot[8 * kW + tid] = 0.f;

  f32x16 acc[RBN][kCB];
  f32x4 bfrag[4][kCB];
  zero_acc(acc);
  uint32_t mw[kMaskWords];   // relu-mask words of the layer whose gradient the running GEMM produces
  {
    const uint32_t* mp = mask + ((slot * kDepth + (kDepth - 1)) * kMlpThreads + tid) * kMaskWords + mask_word0;
#pragma unroll
    for (int w = 0; w < kWordsUsed; ++w) mw[w] = mp[w];
    const int wp = (wave * kCB) * 64;      // wave-uniform f32x4 index into the image (head^T first)
    gemm_prefetch_b<kCB>(wimg, wp, 4 * NHB, 8 * 64, bfrag);
    gemm_lds_packed<RBN, kCB>(arow, wimg, wp, 4 * NHB, 8 * 64, acc, bfrag);
  }
  for (int l = kDepth - 1; l >= 0; --l) {
    // the next GEMM's first weight fragments travel while this wave is in its epilogue
    const int wp = (int)(bwd_layer_off(l > 0 ? l : 1, deg) / 4) + (wave * kCB) * 64;
    if (l > 0) gemm_prefetch_b<kCB>(wimg, wp, 32, 8 * 64, bfrag);
    lds_barrier();  // every wave has consumed the columns this wave is about to rewrite
    int tid_e = tid;
    asm volatile("" : "+v"(tid_e));   // see fwd_tile: keeps the epilogue addresses out of the loops' live set
    const int lane_e = tid_e & 63;
#pragma unroll
    for (int c = 0; c < kCB; ++c) {
      const int col = (wave * kCB + c) * 32 + (lane_e & 31);
      float colsum = 0.f;
#pragma unroll
      for (int r = 0; r < RBN; ++r)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int row = r * 32 + frag_row(reg, lane_e);
          const float v = mask_pop(mw[((r * kCB + c) * 16 + reg) >> 5], acc[r][c][reg]);
          lds[row * kLDA + col] = v;
          colsum += v;
        }
      colsum += __shfl_xor(colsum, 32);
      if (lane_e < 32) db_slot[l * kW + col] = colsum;
    }
    // dz_l leaves for HBM during the GEMM that reads it (TileCopy, below); only dz_0 has no GEMM behind it
    if (l == 0) store_wave_cols<RBN>(lds, dz, row0, M, wave, lane_e);
    lds_barrier();
    if (l == 0) break;
    zero_acc(acc);
    const uint32_t* mp = mask + ((slot * kDepth + (l - 1)) * kMlpThreads + tid_e) * kMaskWords + mask_word0;
#pragma unroll
    for (int w = 0; w < kWordsUsed; ++w) mw[w] = mp[w];     // next layer's mask, fetched under the GEMM
    const TileCopy tc = make_tile_copy<RBN>(lds, dz + (int64_t)l * M * kW, row0, M, wave, lane, true);
    gemm_lds_packed<RBN, kCB, true>(arow, wimg, wp, 32, 8 * 64, acc, bfrag, &tc);
  }
  return 0;
}

template <int NHB, bool SKIP, bool DYN>
__global__ __launch_bounds__(kMlpThreads, kMlpWgPerCu * kMlpWaves / 4) void mlp_bwd_data_kernel(
    const float* __restrict__ pkb, const float* __restrict__ d_raw_rgb,
    const float* __restrict__ d_raw_sigma, const uint32_t* __restrict__ mask, int64_t M, int deg, TileSched ts,
    float* __restrict__ dz, float* __restrict__ dbias_partial, uint8_t* __restrict__ chunk_live,
    unsigned int* __restrict__ tile_counter) {
  // activation-gradient tile; bias gradients leave as one [9][256] partial per tile slot (fixed-order sum over the slots in
  // reduce_jobs_kernel): a tile's sums are the same whichever workgroup computes it, so the schedule below is free
  __shared__ __attribute__((aligned(16))) float lds[kTM * kLDA + kTM / kLiveRows + 4];
  int* __restrict__ nz = reinterpret_cast<int*>(lds + kTM * kLDA);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  uint8_t* const tile_live = reinterpret_cast<uint8_t*>(dbias_partial + mask_slots(M) * 9 * kW);
  const int64_t n_slots = ts.n_full + ts.n_half;
  auto run = [&](int64_t slot) {
    float* db = dbias_partial + slot * 9 * kW;
    if (slot < ts.n_full) {
      const int half = bwd_tile<NHB, kRB, SKIP>(lds, db, tile_live + slot, pkb, d_raw_rgb, d_raw_sigma, mask, M, deg,
                                                slot * kTM, slot, dz, nz, chunk_live, tid, lane, wave);
      if (SKIP && half)
        bwd_tile<NHB, kRB / 2, SKIP>(lds, db, tile_live + slot, pkb, d_raw_rgb, d_raw_sigma, mask, M, deg,
                                     slot * kTM + (half - 1) * (kTM / 2), slot, dz, nz, chunk_live, tid, lane, wave, half - 1);
    } else
      bwd_tile<NHB, kRB / 2, SKIP>(lds, db, tile_live + slot, pkb, d_raw_rgb, d_raw_sigma, mask, M, deg,
                             ts.half_row0 + (slot - ts.n_full) * (kTM / 2), slot, dz, nz, chunk_live, tid, lane, wave);
  };
  if (!DYN) {
    for (int64_t slot = blockIdx.x; slot < ts.n_full; slot += gridDim.x) run(slot);
    for (int64_t h = blockIdx.x; h < ts.n_half; h += gridDim.x) run(ts.n_full + h);
  } else {
    // tiles taken from the device counter (TileTicket above).  In skipping mode a skipped tile costs ~1 % of a live one, so
    // the static stride would leave the workgroups that drew few live tiles idle (measured: the kernel at 0.60 of its dense
    // time with 13 % of the rows live).  Results do not depend on the order (per-slot partials, disjoint dz rows).
    const TileTicket tk{nz + kTM / kLiveRows, tile_counter};
    for (int64_t slot = blockIdx.x; slot < n_slots;) {
      const int ticket = tk.draw(tid);
      run(slot);
      slot = tk.take(tid, ticket);
    }
  }
}

int mlp_bwd_partials(int64_t M) {
  const TileSched ts = tile_sched(M, mlp_grid(M));
  return (int)(ts.n_full + ts.n_half);
}

int launch_mlp_bwd_data(const PxoCfg* cfg, const float* packed_bwd, const float* d_raw_rgb,
                        const float* d_raw_sigma, const uint32_t* mask, int64_t M, float* dz,
                        float* dbias_partial, uint8_t* chunk_live, unsigned int* tile_counter, hipStream_t s,
                        bool counter_is_zero, int flags) {
  if (M == 0) return PXO_OK;
  if (tile_counter && !counter_is_zero && hipMemsetAsync(tile_counter, 0, sizeof(unsigned int), s) != hipSuccess) {
    set_error("mlp_bwd_data: hipMemsetAsync(tile counter) failed");
    return PXO_ERR_HIP;
  }
  if (cfg->mlp_precision == PXO_MLP_BF16X6)
    return launch_mlp_bwd_data_x6(cfg, packed_bwd, d_raw_rgb, d_raw_sigma, mask, M, dz, dbias_partial, chunk_live, tile_counter, s,
                                  (flags & kBiasFromWgrad) == 0);
  if (cfg->mlp_precision != PXO_MLP_F32) { set_error("mlp_bwd_data: mlp_precision bf16x3 is inference-only"); return PXO_ERR_UNSUPPORTED; }
  KernelTimer timer(PXO_PROF_MLP_BWD_DATA, M, s);
  dim3 grid_dim(mlp_grid(M)), block(kMlpThreads);
  const TileSched ts = tile_sched(M, grid_dim.x);
#define PXO_BWD_(NHB_, SKIP_, DYN_)                                                                                       \
  hipLaunchKernelGGL((mlp_bwd_data_kernel<NHB_, SKIP_, DYN_>), grid_dim, block, 0, s, packed_bwd, d_raw_rgb, d_raw_sigma, \
                     mask, M, cfg->sh_deg, ts, dz, dbias_partial, chunk_live, tile_counter)
#define PXO_BWD(NHB_)                                                       \
  do {                                                                      \
    if (chunk_live && tile_counter) PXO_BWD_(NHB_, true, true);             \
    else if (chunk_live) PXO_BWD_(NHB_, true, false);                       \
    else if (tile_counter) PXO_BWD_(NHB_, false, true);                     \
    else PXO_BWD_(NHB_, false, false);                                      \
  } while (0)
  switch (head_blocks(cfg->sh_deg)) {
    case 1: PXO_BWD(1); break;
    case 2: PXO_BWD(2); break;
    default: PXO_BWD(3); break;
  }
#undef PXO_BWD
#undef PXO_BWD_
  return check_launch("mlp_bwd_data");
}

}  // namespace pxo
