// The 256 x 256 weight-gradient products of the NeRF-SH MLP, dW_l = X_l^T dZ_l (Dense_1..7; the reverse-mode wgrad of
// nerf_sh/train.py:116 for the Dense layers of nerf_sh/nerf/model_utils.py:60-94), on the bf16 matrix pipe in the float32-accurate
// split precision of mlp_x6_kernels.hip (PxoCfg.mlp_precision = PXO_MLP_BF16X6): every float32 operand x = x1 + x2 + x3
// exactly (bf16 each), a product = the six partial products of order <= 2^-16 on v_mfma_f32_32x32x16_bf16, float32 accumulation.
//
// Same decomposition as wgrad_kernel<256, 256, 2, 2, ..., NSPLIT = 2>: split-K over row ranges, two 4-wave workgroups per
// range (column halves, partners on one XCD), 16-row chunks (= one live flag of the zero-row skipping pass) double-buffered
// in LDS, one slab per range, the SAME slabs / reduce / grid mapping, so the launcher swaps one launch for the other.
// What differs is the staging: float32 rows come from HBM, are split in registers and land in LDS as bf16 planes with the
// CONTRACTION index (the row) contiguous -- [part][k-block of 8 rows][column] x 16 bytes -- which is what both MFMA operands
// of a "TN" product want: a lane's A fragment is 8 rows of one input feature, its B fragment 8 rows of one output column.
// A thread owns one column of the chunk (16 dword loads, lanes along the row: 256-byte wave loads), so its 8-row groups are
// whole fragments: one ds_write_b128 per (part, k-block), conflict-free both ways.
#include "pxo_common.h"
#include "pxo_x6.h"

namespace pxo {

namespace {
constexpr int kGK = kLiveRows;             // 16 rows per chunk = the K of one MFMA
constexpr int kGThreads = 256;
constexpr int kGHalf = kW / 2;             // output columns of a workgroup
constexpr int kGMaxLive = 2048;            // as wgrad_kernels.hip: live-chunk list of a sparse workgroup
static_assert(kGK == 16, "one v_mfma_f32_32x32x16_bf16 per chunk and block");
}  // namespace

template <bool SPARSE>
__global__ __launch_bounds__(kGThreads, 2) void wgrad_x6_kernel(
    const float* __restrict__ X, const float* __restrict__ dZ, int64_t M, int64_t rows_per_wg, int P,
    float* __restrict__ slab, int n_layers, int64_t layer_stride, const uint8_t* __restrict__ chunk_live) {
  // [buffer][part][k-block][column] fragments of 8 bf16
  __shared__ __attribute__((aligned(16))) bf16x8 xs[2][3][2][kW];
  __shared__ __attribute__((aligned(16))) bf16x8 zs[2][3][2][kGHalf];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;            // wave tile: input features [128 wr, +128) x columns [64 wc, +64) of the half
  int bid = blockIdx.x;
  {
    const int per = gridDim.x / n_layers, g = bid / per;
    bid -= g * per;
    X += (int64_t)g * layer_stride;
    dZ += (int64_t)g * layer_stride;
    slab += (int64_t)g * P * kW * kW;
  }
  const int grp = bid / 16, r16 = bid % 16;
  const int p = grp * 8 + (r16 & 7), half = r16 >> 3;
  if (p >= P) return;
  const int ncol0 = half * kGHalf;
  const int64_t r_begin = (int64_t)p * rows_per_wg;
  int64_t r_end = r_begin + rows_per_wg;
  if (r_end > M) r_end = M;
  const int nchunks = (int)((r_end - r_begin + kGK - 1) / kGK);
  const int64_t range_rows = r_end - r_begin;

  // buffer loads on the workgroup's own row range: rows past its end read as zeros by the bounds check (wgrad_kernels.hip)
  const __amdgpu_buffer_rsrc_t rsX =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X + r_begin * kW), 0, (int)(range_rows * kW * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsZ = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(dZ + r_begin * kW + ncol0), 0, (int)((range_rows * kW - ncol0) * 4), 0x00020000);
  // X: thread = input feature tid, 16 rows; dZ: thread = (column tid & 127, k-block tid >> 7), 8 rows
  const uint32_t xvo = (uint32_t)tid * 4u;
  const int zc = tid & (kGHalf - 1), zkb = tid >> 7;
  const uint32_t zvo = (uint32_t)(zkb * 8 * kW + zc) * 4u;

  struct Stage { float x[kGK]; float z[8]; };
  auto load_chunk = [&](int ch, Stage& st) {
    const int so = ch * (kGK * kW * 4);
#pragma unroll
    for (int r = 0; r < kGK; ++r)
      st.x[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsX, xvo, so + r * (kW * 4), 0));
#pragma unroll
    for (int r = 0; r < 8; ++r)
      st.z[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsZ, zvo, so + r * (kW * 4), 0));
  };
  // 8 rows of one column -> the three 16-byte fragments
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
  auto put_x = [&](int buf, int kbi, const Frag3& f) {
#pragma unroll
    for (int part = 0; part < 3; ++part) xs[buf][part][kbi][tid] = __builtin_bit_cast(bf16x8, f.q[part]);
  };
  auto put_z = [&](int buf, const Frag3& f) {
#pragma unroll
    for (int part = 0; part < 3; ++part) zs[buf][part][zkb][zc] = __builtin_bit_cast(bf16x8, f.q[part]);
  };
  auto store_chunk = [&](int buf, const Stage& st) {
    put_x(buf, 0, split8(st.x));
    put_x(buf, 1, split8(st.x + 8));
    put_z(buf, split8(st.z));
  };

  constexpr int RB = 4, CB = 2;
  f32x16 acc[RB][CB];
#pragma unroll
  for (int r = 0; r < RB; ++r)
#pragma unroll
    for (int c = 0; c < CB; ++c)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[r][c][i] = 0.f;

  const int kb = lane >> 5, l32 = lane & 31;
  // One chunk.  The six products of a block go through a chain that starts from ZERO and is added to the block's accumulator
  // once (16 v_add_f32): every rounding inside the chain is relative to the 16-row partial sum, one rounding per chunk is
  // relative to the running sum (the float32-MFMA kernel: eight).  Measured on 40,000 rows against float64: 0.62 - 0.76 x the
  // float32-MFMA kernel's mean error; the chain run on the accumulator itself is 14 % faster and 1.2 - 1.4 x ITS error.
  // Column-block fragments stay in registers, the four row blocks' fragments stream through.
  // Branch-free (the last chunk re-loads and re-stores itself into the buffer nobody reads): ONE scheduling region per chunk.
  auto run_chunk = [&](int buf, Stage& st, int ld_ch) {
    load_chunk(ld_ch, st);
    bf16x8 b[CB][3];
#pragma unroll
    for (int c = 0; c < CB; ++c)
#pragma unroll
      for (int part = 0; part < 3; ++part) b[c][part] = zs[buf][part][kb][(wc * CB + c) * 32 + l32];
    Frag3 fx0, fx1, fz;
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      bf16x8 a[3];
#pragma unroll
      for (int part = 0; part < 3; ++part) a[part] = xs[buf][part][kb][(wr * RB + r) * 32 + l32];
      // the leading product FIRST, onto zero; then the corrections onto it (the other order -- the big product last, onto
      // the small sum -- measured 2.5 x the float32-MFMA kernel's error, this one 0.7 x; EXPERIMENTS, round 6)
      constexpr int PA[6] = {0, 0, 1, 1, 0, 2}, PB[6] = {0, 1, 0, 1, 2, 0};
      const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      f32x16 t[CB];
#pragma unroll
      for (int c = 0; c < CB; ++c) t[c] = zero;
#pragma unroll
      for (int k = 0; k < 6; ++k)
#pragma unroll
        for (int c = 0; c < CB; ++c) t[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA[k]], b[c][PB[k]], t[c], 0, 0, 0);
      // the next chunk's split, a third per row block from the second one on (the loads were issued a row block earlier)
      if (r == 1) fx0 = split8(st.x);
      if (r == 2) fx1 = split8(st.x + 8);
      if (r == 3) fz = split8(st.z);
#pragma unroll
      for (int c = 0; c < CB; ++c) acc[r][c] += t[c];
    }
    put_x(buf ^ 1, 0, fx0);
    put_x(buf ^ 1, 1, fx1);
    put_z(buf ^ 1, fz);
    __syncthreads();
  };

  // zero-row skipping: the live chunks of the range, in order (wgrad_kernels.hip; same accumulation order over the chunks
  // that contribute => the same bits as the dense walk)
  constexpr int kMaxLive = SPARSE ? kGMaxLive : 1;
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
      store_chunk(0, st);
    }
    __syncthreads();
    for (int i = 0; i < n_run; ++i) run_chunk(i & 1, st, chunk_at(i + 1 < n_run ? i + 1 : i));
  }

  float* out = slab + (int64_t)p * kW * kW;
#pragma unroll
  for (int r = 0; r < RB; ++r)
#pragma unroll
    for (int c = 0; c < CB; ++c) {
      const int n = ncol0 + (wc * CB + c) * 32 + l32;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int i = (wr * RB + r) * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * kb;
        out[(int64_t)i * kW + n] = acc[r][c][reg];
      }
    }
}

// grid / argument conventions of the float32 launch it replaces (launch_mlp_bwd_weights)
void launch_wgrad_main_x6(const float* acts, const float* dz1, int64_t M, int64_t rpw, int P, float* slab, int n_layers,
                          int64_t layer_stride, const uint8_t* chunk_live, hipStream_t s) {
  const dim3 grid(n_layers * ((P + 7) / 8) * 16), block(kGThreads);
  if (chunk_live)
    hipLaunchKernelGGL((wgrad_x6_kernel<true>), grid, block, 0, s, acts, dz1, M, rpw, P, slab, n_layers, layer_stride, chunk_live);
  else
    hipLaunchKernelGGL((wgrad_x6_kernel<false>), grid, block, 0, s, acts, dz1, M, rpw, P, slab, n_layers, layer_stride,
                       (const uint8_t*)nullptr);
}

}  // namespace pxo
