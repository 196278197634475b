// Weight-gradient GEMMs of the NeRF-SH MLP for gfx950: dW_l = X_l^T dZ_l, contraction over the
// M = rays*samples rows (the reverse-mode wgrad of nerf_sh/train.py:116 for the Dense layers of
// nerf_sh/nerf/model_utils.py:60-94).
//
// Split-K: workgroup p owns a contiguous row range, streams 32-row chunks of X and dZ through
// double-buffered LDS (row-major: with mfma_f32_32x32x2f32 the A^T/B fragments of a "TN" GEMM
// are 32 consecutive floats of one LDS row -> conflict-free ds_read_b32), keeps the whole
// KIN x NOUT product in accumulators and writes one slab; a second kernel adds the slabs in a
// fixed order (deterministic, no float atomics).
#include "pxo_common.h"

namespace pxo {

constexpr int kKC = 32;           // rows per staged chunk
constexpr int kWgThreads = 512;   // 8 waves

template <int KIN, int NOUT, int WR, int WC, bool HEAD>
__global__ __launch_bounds__(kWgThreads) void wgrad_kernel(
    const float* __restrict__ X, const float* __restrict__ dZ, const float* __restrict__ d_raw_sigma,
    int C, int64_t M, int64_t rows_per_wg, float* __restrict__ slab) {
  static_assert(WR * WC == 8, "8 waves");
  constexpr int RB = KIN / 32 / WR, CB = NOUT / 32 / WC;
  constexpr int XV = kKC * KIN / 4 / kWgThreads;              // float4 per thread per X chunk
  constexpr int ZV = HEAD ? kKC * NOUT / kWgThreads           // scalars per thread (head)
                          : kKC * NOUT / 4 / kWgThreads;      // float4 per thread
  static_assert(XV >= 1 && ZV >= 1, "tile too small");
  __shared__ __attribute__((aligned(16))) float xs[2][kKC * KIN];
  __shared__ __attribute__((aligned(16))) float zs[2][kKC * NOUT];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave / WC, wc = wave % WC;
  const int64_t r_begin = blockIdx.x * rows_per_wg;
  int64_t r_end = r_begin + rows_per_wg;
  if (r_end > M) r_end = M;
  const int nchunks = (int)((r_end - r_begin + kKC - 1) / kKC);

  f32x4 xr[XV];
  f32x4 zr4[HEAD ? 1 : ZV];
  float zr1[HEAD ? ZV : 1];

  auto load_chunk = [&](int ch) {
    const int64_t r0 = r_begin + (int64_t)ch * kKC;
#pragma unroll
    for (int i = 0; i < XV; ++i) {
      const int idx = tid + kWgThreads * i;
      const int row = idx / (KIN / 4), c4 = idx % (KIN / 4);
      const int64_t grow = r0 + row;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (grow < r_end) v = *reinterpret_cast<const f32x4*>(X + grow * KIN + c4 * 4);
      xr[i] = v;
    }
    if (HEAD) {
#pragma unroll
      for (int i = 0; i < ZV; ++i) {
        const int idx = tid + kWgThreads * i;
        const int row = idx / NOUT, col = idx % NOUT;
        const int64_t grow = r0 + row;
        float v = 0.f;
        if (grow < r_end) {
          if (col < C) v = dZ[grow * C + col];
          else if (col == C) v = d_raw_sigma[grow];
        }
        zr1[i] = v;
      }
    } else {
#pragma unroll
      for (int i = 0; i < ZV; ++i) {
        const int idx = tid + kWgThreads * i;
        const int row = idx / (NOUT / 4), c4 = idx % (NOUT / 4);
        const int64_t grow = r0 + row;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (grow < r_end) v = *reinterpret_cast<const f32x4*>(dZ + grow * NOUT + c4 * 4);
        zr4[i] = v;
      }
    }
  };
  auto store_chunk = [&](int buf) {
#pragma unroll
    for (int i = 0; i < XV; ++i) {
      const int idx = tid + kWgThreads * i;
      *reinterpret_cast<f32x4*>(&xs[buf][idx * 4]) = xr[i];
    }
    if (HEAD) {
#pragma unroll
      for (int i = 0; i < ZV; ++i) zs[buf][tid + kWgThreads * i] = zr1[i];
    } else {
#pragma unroll
      for (int i = 0; i < ZV; ++i) {
        const int idx = tid + kWgThreads * i;
        *reinterpret_cast<f32x4*>(&zs[buf][idx * 4]) = zr4[i];
      }
    }
  };

  f32x16 acc[RB][CB];
#pragma unroll
  for (int r = 0; r < RB; ++r)
#pragma unroll
    for (int c = 0; c < CB; ++c)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[r][c][i] = 0.f;

  if (nchunks > 0) {
    load_chunk(0);
    store_chunk(0);
  }
  __syncthreads();
  for (int ch = 0; ch < nchunks; ++ch) {
    const int buf = ch & 1;
    if (ch + 1 < nchunks) load_chunk(ch + 1);
    const float* xa = &xs[buf][(lane >> 5) * KIN + (wr * RB) * 32 + (lane & 31)];
    const float* zb = &zs[buf][(lane >> 5) * NOUT + (wc * CB) * 32 + (lane & 31)];
#pragma unroll 4
    for (int kk = 0; kk < kKC; kk += 2) {
      float a[RB], b[CB];
#pragma unroll
      for (int r = 0; r < RB; ++r) a[r] = xa[kk * KIN + r * 32];
#pragma unroll
      for (int c = 0; c < CB; ++c) b[c] = zb[kk * NOUT + c * 32];
#pragma unroll
      for (int r = 0; r < RB; ++r)
#pragma unroll
        for (int c = 0; c < CB; ++c)
          acc[r][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r], b[c], acc[r][c], 0, 0, 0);
    }
    if (ch + 1 < nchunks) store_chunk(buf ^ 1);
    __syncthreads();
  }

  float* out = slab + (int64_t)blockIdx.x * KIN * NOUT;
#pragma unroll
  for (int r = 0; r < RB; ++r)
#pragma unroll
    for (int c = 0; c < CB; ++c) {
      const int n = (wc * CB + c) * 32 + (lane & 31);
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int i = (wr * RB + r) * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
        out[(int64_t)i * NOUT + n] = acc[r][c][reg];
      }
    }
}

// dst[i*dst_ld + (n-col0)] = sum_p slab[p][i][n]   for i < rows_valid, col0 <= n < col0+ncols
__global__ void reduce_slab_kernel(const float* __restrict__ slab, int P, int kin, int nout,
                                   int rows_valid, int col0, int ncols, float* __restrict__ dst,
                                   int dst_ld) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows_valid * ncols) return;
  const int i = idx / ncols, n = col0 + idx % ncols;
  const int64_t e = (int64_t)i * nout + n;
  const int64_t stride = (int64_t)kin * nout;
  float s = 0.f;
  for (int p = 0; p < P; ++p) s += slab[p * stride + e];
  dst[(int64_t)i * dst_ld + (n - col0)] = s;
}

// bias gradients: fixed-order sum of the per-workgroup partials written by mlp_bwd_data_kernel.
// block = (layer 0..8, 32-column group); thread (tsub, c) sums tiles == tsub mod 8.
__global__ void reduce_dbias_kernel(const float* __restrict__ partial, int64_t ntiles, int deg,
                                    float* __restrict__ grads) {
  __shared__ float red[8][32];
  const int l = blockIdx.x / 8, cg = blockIdx.x % 8;
  const int c = threadIdx.x & 31, tsub = threadIdx.x >> 5;
  const int col = cg * 32 + c;
  float s = 0.f;
  for (int64_t t = tsub; t < ntiles; t += 8) s += partial[(t * 9 + l) * kW + col];
  red[tsub][c] = s;
  __syncthreads();
  if (tsub == 0) {
    float tot = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) tot += red[i][c];
    if (l < 8) {
      grads[leaf_bias_off(l, deg) + col] = tot;
    } else {
      const int C = rgb_channels(deg);
      if (col < C) grads[leaf_bias_off(9, deg) + col] = tot;
      else if (col == C) grads[leaf_bias_off(8, deg)] = tot;
    }
  }
}

static void split_rows(int64_t M, int64_t* rows_per_wg, int* P) {
  int64_t target = num_cus();
  int64_t rpw = (M + target - 1) / target;
  rpw = (rpw + kKC - 1) / kKC * kKC;
  if (rpw < kKC) rpw = kKC;
  *rows_per_wg = rpw;
  *P = (int)((M + rpw - 1) / rpw);
}

size_t wgrad_workspace_bytes(const PxoCfg* cfg, int64_t M) {
  // one KIN x NOUT (<= 256x256) slab per workgroup; split_rows never makes more than num_cus()
  // workgroups, so the size does not depend on M (two passes of different M share one workspace)
  (void)cfg; (void)M;
  return (size_t)num_cus() * kW * kW * sizeof(float);
}

template <int NHB>
static void launch_head_wgrad(const float* X, const float* d_raw_rgb, const float* d_raw_sigma, int C,
                              int64_t M, int64_t rpw, int P, float* slab, hipStream_t s) {
  hipLaunchKernelGGL((wgrad_kernel<kW, 32 * NHB, 8, 1, true>), dim3(P), dim3(kWgThreads), 0, s, X, d_raw_rgb,
                     d_raw_sigma, C, M, rpw, slab);
}

int launch_mlp_bwd_weights(const PxoCfg* cfg, const float* acts, const float* enc, const float* dz,
                           const float* d_raw_rgb, const float* d_raw_sigma,
                           const float* dbias_partial, int64_t M, float* grads, void* ws,
                           size_t ws_bytes, hipStream_t s) {
  if (M == 0) return PXO_OK;
  const int deg = cfg->sh_deg;
  const int C = rgb_channels(deg);
  int64_t rpw; int P;
  split_rows(M, &rpw, &P);
  if (ws_bytes < (size_t)P * kW * kW * sizeof(float)) {
    set_error("wgrad workspace too small: %zu < %zu", ws_bytes, (size_t)P * kW * kW * sizeof(float));
    return PXO_ERR_WORKSPACE;
  }
  float* slab = reinterpret_cast<float*>(ws);
  const int64_t MW = M * kW;
  auto reduce = [&](int kin, int nout, int rows_valid, int col0, int ncols, float* dst, int dst_ld) {
    const int n = rows_valid * ncols;
    hipLaunchKernelGGL(reduce_slab_kernel, dim3((n + 255) / 256), dim3(256), 0, s, slab, P, kin, nout,
                       rows_valid, col0, ncols, dst, dst_ld);
  };
  // Dense_0: enc^T dz_0  (63 valid input rows)
  hipLaunchKernelGGL((wgrad_kernel<kEncPad, kW, 2, 4, false>), dim3(P), dim3(kWgThreads), 0, s, enc, dz,
                     nullptr, 0, M, rpw, slab);
  reduce(kEncPad, kW, kEnc, 0, kW, grads + leaf_kernel_off(0, deg), kW);
  // Dense_1..7: h_{l-1}^T dz_l  (for l = 5 these are the first 256 input rows)
  for (int l = 1; l < kDepth; ++l) {
    {
    KernelTimer timer(PXO_PROF_WGRAD_MAIN, M, s);
    hipLaunchKernelGGL((wgrad_kernel<kW, kW, 4, 2, false>), dim3(P), dim3(kWgThreads), 0, s,
                       acts + (int64_t)(l - 1) * MW, dz + (int64_t)l * MW, nullptr, 0, M, rpw, slab);
    }
    reduce(kW, kW, kW, 0, kW, grads + leaf_kernel_off(l, deg), kW);
  }
  // Dense_5 skip rows 256..318: enc^T dz_5
  hipLaunchKernelGGL((wgrad_kernel<kEncPad, kW, 2, 4, false>), dim3(P), dim3(kWgThreads), 0, s, enc,
                     dz + (int64_t)5 * MW, nullptr, 0, M, rpw, slab);
  reduce(kEncPad, kW, kEnc, 0, kW, grads + leaf_kernel_off(5, deg) + (int64_t)kW * kW, kW);
  // heads: h7^T [d_raw_rgb | d_raw_sigma]
  const float* h7 = acts + (int64_t)7 * MW;
  const int nhb = head_blocks(deg);
  if (nhb == 1) launch_head_wgrad<1>(h7, d_raw_rgb, d_raw_sigma, C, M, rpw, P, slab, s);
  else if (nhb == 2) launch_head_wgrad<2>(h7, d_raw_rgb, d_raw_sigma, C, M, rpw, P, slab, s);
  else launch_head_wgrad<3>(h7, d_raw_rgb, d_raw_sigma, C, M, rpw, P, slab, s);
  reduce(kW, 32 * nhb, kW, 0, C, grads + leaf_kernel_off(9, deg), C);
  reduce(kW, 32 * nhb, kW, C, 1, grads + leaf_kernel_off(8, deg), 1);
  // biases
  hipLaunchKernelGGL(reduce_dbias_kernel, dim3(9 * 8), dim3(256), 0, s, dbias_partial, (int64_t)mlp_bwd_partials(M), deg, grads);
  return check_launch("mlp_bwd_weights");
}

}  // namespace pxo
