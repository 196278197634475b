// The 256 x 256 weight-gradient products of the NeRF-SH MLP, dW_l = X_l^T dZ_l (Dense_1..7; the reverse-mode wgrad of
// nerf_sh/train.py:116 for the Dense layers of nerf_sh/nerf/model_utils.py:60-94), on the bf16 matrix pipe in the float32-accurate
// split precision of mlp_x6_kernels.hip (PxoCfg.mlp_precision = PXO_MLP_BF16X6): every float32 operand x = x1 + x2 + x3
// exactly (bf16 each), a product = the six partial products of order <= 2^-16 on v_mfma_f32_32x32x16_bf16, float32 accumulation.
//
// Split-K over the row ranges of wgrad_kernel<256, 256, ...> (same ranges, same slabs, same fixed-order reduce, same live-chunk
// walk in the zero-row skipping pass), but ONE 8-wave workgroup per range computes the whole 256 x 256 product (wave tile
// 128 x 64: 8 accumulator blocks), so every operand element is split once.  16-row chunks (= one live flag, = the K of one
// MFMA) double-buffered in LDS.  What differs from the float32 kernel is the staging: float32 rows come from HBM, are split in
// registers and land in LDS as bf16 planes with the CONTRACTION index (the row) contiguous -- [part][k-block of 8 rows][column]
// x 16 bytes -- which is what both MFMA operands of a "TN" product want: a lane's A fragment is 8 rows of one input feature,
// its B fragment 8 rows of one output column.  A thread owns (one column, one k-block) of X and of dZ: 8 + 8 dword loads with
// the lanes along the row (256-byte wave loads), each group of 8 a whole fragment: one ds_write_b128 per part, conflict-free
// both ways.  While it streams dZ the kernel also sums its columns per range: the bias gradients of Dense_1..7 (kBiasFromWgrad in
// pxo_common.h; backward(data) then drops its lane reductions for those layers, 6 % of its time).  (A first version with two 4-wave workgroups per range -- column halves, X split twice -- gave the same bits
// and was 7 % slower: 2.63 vs 2.44 ms per launch.)
#include "pxo_common.h"
#include "pxo_x6.h"

namespace pxo {

namespace {
constexpr int kGK = kLiveRows;             // 16 rows per chunk = the K of one MFMA
constexpr int kGThreads = 512;
static_assert(kGK == 16, "one v_mfma_f32_32x32x16_bf16 per chunk and block");
}  // namespace

template <bool SPARSE>
__global__ __launch_bounds__(kGThreads, 2) void wgrad_x6_kernel(
    const float* __restrict__ X, const float* __restrict__ dZ, int64_t M, int64_t rows_per_wg, int P,
    float* __restrict__ slab, int n_layers, int64_t layer_stride, const uint8_t* __restrict__ chunk_live,
    float* __restrict__ dz_colsum) {
  __shared__ __attribute__((aligned(16))) bf16x8 xs[2][3][2][kW];
  __shared__ __attribute__((aligned(16))) bf16x8 zs[2][3][2][kW];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;            // wave tile: input features [128 wr, +128) x output columns [64 wc, +64)
  int p = blockIdx.x;
  {
    const int per = gridDim.x / n_layers, g = p / per;
    p -= g * per;
    X += (int64_t)g * layer_stride;
    dZ += (int64_t)g * layer_stride;
    slab += (int64_t)g * P * kW * kW;
    dz_colsum += (int64_t)g * P * kW;
  }
  if (p >= P) return;
  const int64_t r_begin = (int64_t)p * rows_per_wg;
  int64_t r_end = r_begin + rows_per_wg;
  if (r_end > M) r_end = M;
  const int nchunks = (int)((r_end - r_begin + kGK - 1) / kGK);
  const int64_t range_rows = r_end - r_begin;
  const __amdgpu_buffer_rsrc_t rsX =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X + r_begin * kW), 0, (int)(range_rows * kW * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsZ =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dZ + r_begin * kW), 0, (int)(range_rows * kW * 4), 0x00020000);
  // thread = (column tid & 255, k-block tid >> 8): 8 rows of X and 8 rows of dZ
  const int sc = tid & (kW - 1), skb = tid >> 8;
  const uint32_t svo = (uint32_t)(skb * 8 * kW + sc) * 4u;

  struct Stage { float x[8]; float z[8]; };
  auto load_chunk = [&](int ch, Stage& st) {
    const int so = ch * (kGK * kW * 4);
#pragma unroll
    for (int r = 0; r < 8; ++r)
      st.x[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsX, svo, so + r * (kW * 4), 0));
#pragma unroll
    for (int r = 0; r < 8; ++r)
      st.z[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsZ, svo, so + r * (kW * 4), 0));
  };
  struct Frag3 { u32x4 q[3]; };
  auto split8 = [&](const float* v) {
    Frag3 f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      uint32_t p1, p2, p3;
      split3_pair(v[2 * t], v[2 * t + 1], p1, p2, p3);
      f.q[0][t] = p1; f.q[1][t] = p2; f.q[2][t] = p3;
    }
    return f;
  };
  auto put = [&](bf16x8 (*dst)[2][kW], const Frag3& f) {
#pragma unroll
    for (int part = 0; part < 3; ++part) dst[part][skb][sc] = __builtin_bit_cast(bf16x8, f.q[part]);
  };

  constexpr int RB = 4, CB = 2;
  f32x16 acc[RB][CB];
#pragma unroll
  for (int r = 0; r < RB; ++r)
#pragma unroll
    for (int c = 0; c < CB; ++c)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[r][c][i] = 0.f;

  // column sums of dZ over the range (= this range's share of the layer's bias gradient): this thread's 8 rows of every chunk,
  // in chunk order; the two k-blocks of a column meet in LDS at the end
  float zsum = 0.f;
  // (row order, one accumulator, `on` = 1 or 0 as a FACTOR: no temporaries and no branch -- the kernel sits at its register limit
  // and a chunk is one scheduling region)
  auto colsum8 = [&](const float* v, float on) {
#pragma unroll
    for (int i = 0; i < 8; ++i) zsum = __builtin_fmaf(on, v[i], zsum);
  };
  const int kb = lane >> 5, l32 = lane & 31;
  // chunk `buf` is multiplied; the loads of chunk + 1 are issued first and split / stored to the other buffer behind the MFMAs
  auto run_chunk = [&](int buf, Stage& st, int ld_ch, float more) {
    load_chunk(ld_ch, st);
    bf16x8 b[CB][3];
#pragma unroll
    for (int c = 0; c < CB; ++c)
#pragma unroll
      for (int part = 0; part < 3; ++part) b[c][part] = zs[buf][part][kb][(wc * CB + c) * 32 + l32];
    bf16x8 an[3];
#pragma unroll
    for (int part = 0; part < 3; ++part) an[part] = xs[buf][part][kb][(wr * RB) * 32 + l32];
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      bf16x8 a[3];
#pragma unroll
      for (int part = 0; part < 3; ++part) a[part] = an[part];
      if (r + 1 < RB) {                        // the next row block's fragments are requested before this block's MFMAs
#pragma unroll
        for (int part = 0; part < 3; ++part) an[part] = xs[buf][part][kb][(wr * RB + r + 1) * 32 + l32];
        __builtin_amdgcn_sched_barrier(0);
      }
      // the leading product FIRST, onto zero; then the corrections onto it (the other order -- the big product last, onto the
      // small sum -- measured 2.5 x the float32-MFMA kernel's error, this one 0.7 x; EXPERIMENTS, round 6)
      constexpr int PA[6] = {0, 0, 1, 1, 0, 2}, PB[6] = {0, 1, 0, 1, 2, 0};
      const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      f32x16 t[CB];
#pragma unroll
      for (int c = 0; c < CB; ++c) t[c] = zero;
#pragma unroll
      for (int k = 0; k < 6; ++k)
#pragma unroll
        for (int c = 0; c < CB; ++c) t[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA[k]], b[c][PB[k]], t[c], 0, 0, 0);
      // the next chunk's split: X behind the second row block's MFMAs, dZ behind the third's (measured, ms per launch: one block
      // earlier 2.42 -- the loads have not landed --, one later 2.37, at the end 2.38, this 2.29)
      if (r == 1) put(xs[buf ^ 1], split8(st.x));
      if (r == 2) {
        put(zs[buf ^ 1], split8(st.z));
        colsum8(st.z, more);                   // (past the end the last chunk is staged a second time: not summed again)
      }
#pragma unroll
      for (int c = 0; c < CB; ++c) acc[r][c] += t[c];
    }
    __syncthreads();
  };

  constexpr int kMaxLive = SPARSE ? kMaxLiveChunks : 1;
  __shared__ uint16_t live_list[kMaxLive];
  __shared__ int live_count;
  const bool sparse = SPARSE && chunk_live != nullptr && nchunks <= kMaxLive;
  int n_run = nchunks;
  if (SPARSE && sparse) {
    if (wave == 0) {
      const uint8_t* fl = chunk_live + r_begin / kGK;
      int base = 0;
      for (int c0 = 0; c0 < nchunks; c0 += 64) {
        const int c = c0 + lane;
        const bool lv = c < nchunks && fl[c] != 0;
        const uint64_t m = __builtin_amdgcn_ballot_w64(lv);
        if (lv) live_list[base + __builtin_popcountll(m & ((1ull << lane) - 1ull))] = (uint16_t)c;
        base += __builtin_popcountll(m);
      }
      if (lane == 0) live_count = base;
    }
    __syncthreads();
    n_run = live_count;
  }
  auto chunk_at = [&](int i) -> int {
    if (SPARSE && sparse) return __builtin_amdgcn_readfirstlane((int)live_list[i]);
    return i;
  };
  {
    Stage st;
    if (n_run > 0) {
      load_chunk(chunk_at(0), st);
      put(xs[0], split8(st.x));
      put(zs[0], split8(st.z));
      colsum8(st.z, 1.f);
    }
    __syncthreads();
    for (int i = 0; i < n_run; ++i) run_chunk(i & 1, st, chunk_at(i + 1 < n_run ? i + 1 : i), i + 1 < n_run ? 1.f : 0.f);
  }

  // the range's column sums: the two k-block partials of a column through LDS (the chunk buffers are free after the last barrier)
  {
    float* zsum_lds = reinterpret_cast<float*>(&xs[0][0][0][0]);
    if (skb == 1) zsum_lds[sc] = zsum;
    __syncthreads();
    if (skb == 0) dz_colsum[(int64_t)p * kW + sc] = zsum + zsum_lds[sc];
  }
  float* out = slab + (int64_t)p * kW * kW;
#pragma unroll
  for (int r = 0; r < RB; ++r)
#pragma unroll
    for (int c = 0; c < CB; ++c) {
      const int n = (wc * CB + c) * 32 + l32;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int i = (wr * RB + r) * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * kb;
        out[(int64_t)i * kW + n] = acc[r][c][reg];
      }
    }
}

// grid / argument conventions of the float32 launch it replaces (launch_mlp_bwd_weights)
void launch_wgrad_main_x6(const float* acts, const float* dz1, int64_t M, int64_t rpw, int P, float* slab, int n_layers,
                          int64_t layer_stride, const uint8_t* chunk_live, float* dz_colsum, hipStream_t s) {
  const dim3 grid(n_layers * P), block(kGThreads);
  if (chunk_live)
    hipLaunchKernelGGL((wgrad_x6_kernel<true>), grid, block, 0, s, acts, dz1, M, rpw, P, slab, n_layers, layer_stride, chunk_live, dz_colsum);
  else
    hipLaunchKernelGGL((wgrad_x6_kernel<false>), grid, block, 0, s, acts, dz1, M, rpw, P, slab, n_layers, layer_stride,
                       (const uint8_t*)nullptr, dz_colsum);
}

}  // namespace pxo
