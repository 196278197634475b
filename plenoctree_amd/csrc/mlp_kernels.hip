// Fused positional-encoding + NeRF-SH MLP forward / backward(data) for gfx950 (MI355X).
//
// Replaces MLP.__call__ of nerf_sh/nerf/model_utils.py:43-94 (torch twin
// octree/nerf/model_utils.py:87-158) and posenc (:145-173) for the use_viewdirs=false /
// SH configuration (nerf_sh/config/blender.yaml, tt.yaml), plus its reverse-mode data path.
//
// Design (exact f32, v_mfma_f32_32x32x2_f32):
//  * one workgroup = 128 samples; the 128x256 activation tile lives in LDS (row stride 260
//    floats: conflict-free ds_read_b128 A-fragments) for all 8 layers and is updated in place;
//  * weights are pre-packed in MFMA fragment order (pxo_common.h packed_index) so the B operand
//    is one coalesced 16 B/lane load straight from L2 into registers -- each wave owns a
//    disjoint 64-column slice of the layer, so weights need no LDS staging at all;
//  * the K order inside a dot product is permuted (lane half h, sub-step j -> k = 8g+4h+j) so
//    that one ds_read_b128 / one 16 B global load feeds four consecutive MFMAs;
//  * post-ReLU activations are streamed to HBM once (for the weight-gradient GEMMs) together
//    with a 1-bit relu mask in fragment order, so the backward-data kernel never re-reads them.
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

// writes posenc of the tile's 128 points into lds[:, 0:64]
template <int NT>
__device__ __forceinline__ void posenc_tile(float* __restrict__ lds, const float* __restrict__ pts,
                                            const GridSpec& grid, int64_t row0, int64_t M, int tid) {
  constexpr int kParts = NT / kTM;                // 2 or 4
  constexpr int kColsPer = kEncPad / kParts;      // 32 or 16
  const int row = tid % kTM, part = tid / kTM;
  const int64_t grow = row0 + row;
  float p0 = 0.f, p1 = 0.f, p2 = 0.f;
  if (grow < M) {
    if (grid.enabled) grid_point(grid, grow, p0, p1, p2);
    else { p0 = pts[grow * 3]; p1 = pts[grow * 3 + 1]; p2 = pts[grow * 3 + 2]; }
  }
#pragma unroll 4
  for (int i = 0; i < kColsPer; ++i) {
    const int col = part * kColsPer + i;
    lds[row * kLDA + col] = enc_value(p0, p1, p2, col);
  }
}

// ------------------------------------------------------------------------------------------
// the 128-row x (CBN*32)-col wave GEMM: A from LDS (ds_read_b128), B from the packed image
// ------------------------------------------------------------------------------------------
template <int RBN, int CBN>
__device__ __forceinline__ void gemm_lds_packed(const float* __restrict__ arow,
                                                const f32x4* __restrict__ wp, int kgroups,
                                                int kg_stride, f32x16 (&acc)[RBN][CBN]) {
  f32x4 a[RBN], b[CBN];
#pragma unroll
  for (int r = 0; r < RBN; ++r) a[r] = *reinterpret_cast<const f32x4*>(arow + r * 32 * kLDA);
#pragma unroll
  for (int c = 0; c < CBN; ++c) b[c] = wp[c * 64];
  for (int g = 0; g < kgroups; ++g) {
    const int gn = (g + 1 < kgroups) ? g + 1 : g;
    f32x4 an[RBN], bn[CBN];
#pragma unroll
    for (int c = 0; c < CBN; ++c) bn[c] = wp[(int64_t)gn * kg_stride + c * 64];
#pragma unroll
    for (int r = 0; r < RBN; ++r) an[r] = *reinterpret_cast<const f32x4*>(arow + r * 32 * kLDA + gn * 8);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < RBN; ++r)
#pragma unroll
        for (int c = 0; c < CBN; ++c)
          acc[r][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r][j], b[c][j], acc[r][c], 0, 0, 0);
#pragma unroll
    for (int r = 0; r < RBN; ++r) a[r] = an[r];
#pragma unroll
    for (int c = 0; c < CBN; ++c) b[c] = bn[c];
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

// ------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------
// NW = waves per workgroup: 4 (one per SIMD, each 128 rows x 64 cols) or 8 (two per SIMD, each
// 128 rows x 32 cols; the second wave's MFMAs cover the first one's waits and epilogues).
template <int NHB, bool SAVE, int NW>
__global__ __launch_bounds__(NW * 64, NW / 4) void mlp_fwd_kernel(
    const float* __restrict__ pk, const float* __restrict__ pts, GridSpec grid, int64_t M, int deg,
    float* __restrict__ raw_rgb, float* __restrict__ raw_sigma, float* __restrict__ acts,
    float* __restrict__ enc_out, uint32_t* __restrict__ mask) {
  constexpr int NT = NW * 64, CPW = 8 / NW, MW = 4 * CPW * 16 / 32;
  __shared__ __attribute__((aligned(16))) float lds[kTM * kLDA];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t tile = blockIdx.x;
  const int64_t row0 = tile * kTM;
  const bool full = row0 + kTM <= M;
  const int C = rgb_channels(deg);
  const float* __restrict__ bias = pk + fwd_bias_off(deg);

  posenc_tile<NT>(lds, pts, grid, row0, M, tid);
  __syncthreads();
  if (SAVE) {  // coalesced copy of the encoded tile (layer-0 / layer-5 weight gradients)
#pragma unroll
    for (int i = 0; i < kTM * kEncPad / 4 / NT; ++i) {
      const int idx = tid + NT * i;
      const int row = idx >> 4, c4 = idx & 15;
      if (row0 + row < M)
        *reinterpret_cast<f32x4*>(enc_out + (row0 + row) * kEncPad + c4 * 4) =
            *reinterpret_cast<const f32x4*>(lds + row * kLDA + c4 * 4);
    }
  }

  const float* arow = lds + (lane & 31) * kLDA + (lane >> 5) * 4;
  f32x16 acc[4][CPW];
  for (int l = 0; l < kDepth; ++l) {
    zero_acc(acc);
    const f32x4* wp = reinterpret_cast<const f32x4*>(pk + fwd_layer_off(l)) + (wave * CPW) * 64 + lane;
    gemm_lds_packed<4, CPW>(arow, wp, l == 0 ? 8 : 32, 8 * 64, acc);
    if (l == 5) {
      // skip connection (model_utils.py:70-71): x = concat([h4, inputs]) -> the 64 encoded
      // columns are a second K segment; the encoding is recomputed into the consumed tile.
      __syncthreads();
      posenc_tile<NT>(lds, pts, grid, row0, M, tid);
      __syncthreads();
      gemm_lds_packed<4, CPW>(arow, wp + (int64_t)32 * 8 * 64, 8, 8 * 64, acc);
    }
    __syncthreads();  // every wave has consumed the input tile
    uint32_t mw[MW];
#pragma unroll
    for (int w = 0; w < MW; ++w) mw[w] = 0u;
    float* __restrict__ act_l = SAVE ? acts + (int64_t)l * M * kW + row0 * kW : nullptr;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < CPW; ++c) {
        const int col = (wave * CPW + c) * 32 + (lane & 31);
        const float b = bias[l * kW + col];
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int row = r * 32 + frag_row(reg, lane);
          const float v = fmaxf(acc[r][c][reg] + b, 0.f);
          lds[row * kLDA + col] = v;
          if (SAVE) {
            const int bit = (r * CPW + c) * 16 + reg;
            if (v > 0.f) mw[bit >> 5] |= 1u << (bit & 31);
            if (full || row0 + row < M) act_l[row * kW + col] = v;
          }
        }
      }
    if (SAVE) {
      uint32_t* mp = mask + ((tile * kDepth + l) * NT + tid) * MW;
#pragma unroll
      for (int w = 0; w < MW; ++w) mp[w] = mw[w];
    }
    __syncthreads();
  }

  // heads: [raw_rgb | raw_sigma] = h7 @ [Dense_9 | Dense_8] + b (model_utils.py:72-74, :91-93);
  // wave w owns row block w%4 and the column blocks (w/4), (w/4)+NW/4, ...
  {
    constexpr int CSTEP = NW / 4;                       // 1 or 2
    constexpr int HMAX = (NHB + CSTEP - 1) / CSTEP;     // column blocks per wave (upper bound)
    const int rb = wave & 3, cb0 = wave >> 2;
    f32x16 hacc[HMAX];
#pragma unroll
    for (int i = 0; i < HMAX; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) hacc[i][e] = 0.f;
    const f32x4* wp = reinterpret_cast<const f32x4*>(pk + fwd_layer_off(8)) + lane;
    const float* ar = arow + rb * 32 * kLDA;
    for (int g = 0; g < 32; ++g) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(ar + g * 8);
#pragma unroll
      for (int i = 0; i < HMAX; ++i) {
        const int cb = cb0 + i * CSTEP;
        if (cb < NHB) {
          const f32x4 b = wp[((int64_t)g * NHB + cb) * 64];
#pragma unroll
          for (int j = 0; j < 4; ++j) hacc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], hacc[i], 0, 0, 0);
        }
      }
    }
    const float* hb = bias + 8 * kW;
#pragma unroll
    for (int i = 0; i < HMAX; ++i) {
      const int cb = cb0 + i * CSTEP;
      if (cb < NHB) {
        const int col = cb * 32 + (lane & 31);
        const float b = hb[col];
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int64_t grow = row0 + rb * 32 + frag_row(reg, lane);
          if (grow < M) {
            const float v = hacc[i][reg] + b;
            if (col < C) { if (raw_rgb) raw_rgb[grow * C + col] = v; }
            else if (col == C) raw_sigma[grow] = v;
          }
        }
      }
    }
  }
}

int g_mlp_waves = 8;   // pxo_set_option("mlp_waves", 4|8)

template <int NHB, int NW>
static void launch_fwd_nw(const float* pk, const float* pts, const GridSpec& grid, int64_t M, int deg,
                          float* raw_rgb, float* raw_sigma, float* acts, float* enc, uint32_t* mask,
                          hipStream_t s) {
  dim3 grid_dim((unsigned)num_tiles(M)), block(NW * 64);
  if (acts)
    hipLaunchKernelGGL((mlp_fwd_kernel<NHB, true, NW>), grid_dim, block, 0, s, pk, pts, grid, M, deg,
                       raw_rgb, raw_sigma, acts, enc, mask);
  else
    hipLaunchKernelGGL((mlp_fwd_kernel<NHB, false, NW>), grid_dim, block, 0, s, pk, pts, grid, M, deg,
                       raw_rgb, raw_sigma, acts, enc, mask);
}

template <int NHB>
static int launch_fwd_nhb(const PxoCfg* cfg, const float* pk, const float* pts, const GridSpec& grid,
                          int64_t M, float* raw_rgb, float* raw_sigma, float* acts, float* enc,
                          uint32_t* mask, hipStream_t s) {
  KernelTimer timer(PXO_PROF_MLP_FWD, M, s);
  if (g_mlp_waves == 8)
    launch_fwd_nw<NHB, 8>(pk, pts, grid, M, cfg->sh_deg, raw_rgb, raw_sigma, acts, enc, mask, s);
  else
    launch_fwd_nw<NHB, 4>(pk, pts, grid, M, cfg->sh_deg, raw_rgb, raw_sigma, acts, enc, mask, s);
  return check_launch("mlp_fwd");
}

static int launch_fwd_any(const PxoCfg* cfg, const float* pk, const float* pts, const GridSpec& grid,
                          int64_t M, float* raw_rgb, float* raw_sigma, float* acts, float* enc,
                          uint32_t* mask, hipStream_t s) {
  if (M == 0) return PXO_OK;
  switch (head_blocks(cfg->sh_deg)) {
    case 1: return launch_fwd_nhb<1>(cfg, pk, pts, grid, M, raw_rgb, raw_sigma, acts, enc, mask, s);
    case 2: return launch_fwd_nhb<2>(cfg, pk, pts, grid, M, raw_rgb, raw_sigma, acts, enc, mask, s);
    default: return launch_fwd_nhb<3>(cfg, pk, pts, grid, M, raw_rgb, raw_sigma, acts, enc, mask, s);
  }
}

int launch_mlp_fwd(const PxoCfg* cfg, const float* packed_fwd, const float* pts, int64_t M,
                   float* raw_rgb, float* raw_sigma, float* acts, float* enc, uint32_t* mask,
                   hipStream_t s) {
  GridSpec g;
  g.enabled = 0; g.reso = 1; g.x0 = 0;
  for (int i = 0; i < 3; ++i) { g.off[i] = 0.f; g.scale[i] = 1.f; }
  return launch_fwd_any(cfg, packed_fwd, pts, g, M, raw_rgb, raw_sigma, acts, enc, mask, s);
}

int launch_mlp_fwd_grid(const PxoCfg* cfg, const float* packed_fwd, int reso, int x0, int x1,
                        const float* off, const float* scale, float* sigma_out, hipStream_t s) {
  GridSpec g;
  g.enabled = 1; g.reso = reso; g.x0 = x0;
  for (int i = 0; i < 3; ++i) { g.off[i] = off[i]; g.scale[i] = scale[i]; }
  const int64_t M = (int64_t)(x1 - x0) * reso * reso;
  return launch_fwd_any(cfg, packed_fwd, nullptr, g, M, nullptr, sigma_out, nullptr, nullptr, nullptr, s);
}

// ------------------------------------------------------------------------------------------
// backward (data): d_raw -> dz_7 .. dz_0
// ------------------------------------------------------------------------------------------
template <int NHB, int NW>
__global__ __launch_bounds__(NW * 64, NW / 4) void mlp_bwd_data_kernel(
    const float* __restrict__ pkb, const float* __restrict__ d_raw_rgb,
    const float* __restrict__ d_raw_sigma, const uint32_t* __restrict__ mask, int64_t M, int deg,
    float* __restrict__ dz, float* __restrict__ dbias_partial) {
  constexpr int NT = NW * 64, CPW = 8 / NW, MW = 4 * CPW * 16 / 32;
  __shared__ __attribute__((aligned(16))) float lds[kTM * kLDA];
  constexpr int NH = 32 * NHB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t tile = blockIdx.x;
  const int64_t row0 = tile * kTM;
  const bool full = row0 + kTM <= M;
  const int C = rgb_channels(deg);

  // d_raw tile -> lds[:, 0:NH] with the head's column order
  for (int idx = tid; idx < kTM * NH; idx += NT) {
    const int row = idx / NH, col = idx - row * NH;
    const int64_t grow = row0 + row;
    float v = 0.f;
    if (grow < M) {
      if (col < C) v = d_raw_rgb[grow * C + col];
      else if (col == C) v = d_raw_sigma[grow];
    }
    lds[row * kLDA + col] = v;
  }
  __syncthreads();
  if (tid < NH) {  // head bias gradient partial of this tile
    float sum = 0.f;
    for (int row = 0; row < kTM; ++row) sum += lds[row * kLDA + tid];
    dbias_partial[(tile * 9 + 8) * kW + tid] = sum;
  }

  const float* arow = lds + (lane & 31) * kLDA + (lane >> 5) * 4;
  f32x16 acc[4][CPW];
  zero_acc(acc);
  {
    const f32x4* wp = reinterpret_cast<const f32x4*>(pkb) + (wave * CPW) * 64 + lane;
    gemm_lds_packed<4, CPW>(arow, wp, 4 * NHB, 8 * 64, acc);
  }
  for (int l = kDepth - 1; l >= 0; --l) {
    uint32_t mw[MW];
    const uint32_t* mp = mask + ((tile * kDepth + l) * NT + tid) * MW;
#pragma unroll
    for (int w = 0; w < MW; ++w) mw[w] = mp[w];
    __syncthreads();  // previous GEMM has consumed the tile
    float* __restrict__ dz_l = dz + (int64_t)l * M * kW + row0 * kW;
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
      const int col = (wave * CPW + c) * 32 + (lane & 31);
      float colsum = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int row = r * 32 + frag_row(reg, lane);
          const int bit = (r * CPW + c) * 16 + reg;
          const float v = ((mw[bit >> 5] >> (bit & 31)) & 1u) ? acc[r][c][reg] : 0.f;
          lds[row * kLDA + col] = v;
          if (full || row0 + row < M) dz_l[row * kW + col] = v;
          colsum += v;
        }
      colsum += __shfl_xor(colsum, 32);
      if (lane < 32) dbias_partial[(tile * 9 + l) * kW + col] = colsum;
    }
    __syncthreads();
    if (l > 0) {
      zero_acc(acc);
      const f32x4* wp = reinterpret_cast<const f32x4*>(pkb + bwd_layer_off(l, deg)) + (wave * CPW) * 64 + lane;
      gemm_lds_packed<4, CPW>(arow, wp, 32, 8 * 64, acc);
    }
  }
}

template <int NHB>
static void launch_bwd_nhb(const float* packed_bwd, const float* d_raw_rgb, const float* d_raw_sigma,
                           const uint32_t* mask, int64_t M, int deg, float* dz, float* dbias_partial,
                           hipStream_t s) {
  dim3 grid_dim((unsigned)num_tiles(M));
  if (g_mlp_waves == 8)
    hipLaunchKernelGGL((mlp_bwd_data_kernel<NHB, 8>), grid_dim, dim3(512), 0, s, packed_bwd, d_raw_rgb,
                       d_raw_sigma, mask, M, deg, dz, dbias_partial);
  else
    hipLaunchKernelGGL((mlp_bwd_data_kernel<NHB, 4>), grid_dim, dim3(256), 0, s, packed_bwd, d_raw_rgb,
                       d_raw_sigma, mask, M, deg, dz, dbias_partial);
}

int launch_mlp_bwd_data(const PxoCfg* cfg, const float* packed_bwd, const float* d_raw_rgb,
                        const float* d_raw_sigma, const uint32_t* mask, int64_t M, float* dz,
                        float* dbias_partial, hipStream_t s) {
  if (M == 0) return PXO_OK;
  KernelTimer timer(PXO_PROF_MLP_BWD_DATA, M, s);
  switch (head_blocks(cfg->sh_deg)) {
    case 1: launch_bwd_nhb<1>(packed_bwd, d_raw_rgb, d_raw_sigma, mask, M, cfg->sh_deg, dz, dbias_partial, s); break;
    case 2: launch_bwd_nhb<2>(packed_bwd, d_raw_rgb, d_raw_sigma, mask, M, cfg->sh_deg, dz, dbias_partial, s); break;
    default: launch_bwd_nhb<3>(packed_bwd, d_raw_rgb, d_raw_sigma, mask, M, cfg->sh_deg, dz, dbias_partial, s); break;
  }
  return check_launch("mlp_bwd_data");
}

}  // namespace pxo
