// Weight-gradient GEMMs of the NeRF-SH MLP for gfx950: dW_l = X_l^T dZ_l, contraction over the
// M = rays*samples rows (the reverse-mode wgrad of nerf_sh/train.py:116 for the Dense layers of
// nerf_sh/nerf/model_utils.py:60-94).
//
// Split-K: workgroup p owns a contiguous row range, streams 32-row chunks of X and dZ through
// double-buffered LDS (row-major: with mfma_f32_32x32x2f32 the A^T/B fragments of a "TN" GEMM
// are 32 consecutive floats of one LDS row -> conflict-free ds_read_b32), keeps the whole
// KIN x NOUT product in accumulators and writes one slab; a second kernel adds the slabs in a
// fixed order (deterministic, no float atomics).
#include <cstdlib>

#include "pxo_common.h"

namespace pxo {

constexpr int kKC = 32;           // row granularity of the split (rows_per_wg is a multiple of it)


// Geometry: NT threads (WR x WC waves), KCH rows per staged chunk, the NOUT columns split over
// NSPLIT workgroups (each owns NOUT/NSPLIT output columns and re-reads X; the NSPLIT partners of a
// row range are placed 8 blocks apart = on the same XCD so the second read of X hits L2).
// DUAL: the NOUT = 512 columns are two row-major [M,256] arrays side by side (dZ | dZ2): Dense_0 and the skip rows of
// Dense_5 share X = enc (model_utils.py:70-71), so one pass over enc produces both gradients.
template <int KIN, int NOUT, int WR, int WC, bool HEAD, int NT, int KCH, int NSPLIT, bool DUAL = false, int SCHED = 0,
          bool SPARSE = false>
__global__ __launch_bounds__(NT) void wgrad_kernel(
    const float* __restrict__ X, const float* __restrict__ dZ, const float* __restrict__ d_raw_sigma,
    int C, int64_t M, int64_t rows_per_wg, int P, float* __restrict__ slab, const float* __restrict__ dZ2 = nullptr,
    int n_layers = 1, int64_t layer_stride = 0, const uint8_t* __restrict__ chunk_live = nullptr) {
  static_assert(WR * WC * 64 == NT, "wave grid");
  static_assert(KCH == kLiveRows, "one live flag per staged chunk");
  static_assert(!DUAL || (NOUT == 2 * kW && NSPLIT == 1 && !HEAD), "dual source: two 256-wide arrays, no split");
  constexpr int NTILE = NOUT / NSPLIT;
  constexpr int RB = KIN / 32 / WR, CB = NTILE / 32 / WC;
  constexpr int XV = KCH * KIN / 4 / NT;                      // float4 per thread per X chunk
  constexpr int ZV = HEAD ? KCH * NTILE / NT                  // scalars per thread (head)
                          : KCH * NTILE / 4 / NT;             // float4 per thread
  static_assert(XV >= 1 && ZV >= 1 && RB >= 1 && CB >= 1, "tile too small");
  static_assert(!HEAD || NSPLIT == 1, "head is not split");
  __shared__ __attribute__((aligned(16))) float xs[2][KCH * KIN];
  __shared__ __attribute__((aligned(16))) float zs[2][KCH * NTILE];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / WC, wc = wave % WC;
  // n_layers > 1: the grid is n_layers equal groups of blocks; group g computes the product of the arrays
  // X + g layer_stride, dZ + g layer_stride into its own P slabs (the 256x256 products of Dense_1..7 in ONE launch)
  int bid = blockIdx.x;
  if (n_layers > 1) {
    const int per = gridDim.x / n_layers, g = bid / per;
    bid -= g * per;
    X += (int64_t)g * layer_stride;
    dZ += (int64_t)g * layer_stride;
    slab += (int64_t)g * P * KIN * NOUT;
  }
  int p, half;
  if (NSPLIT == 1) { p = bid; half = 0; }
  else {
    const int grp = bid / (8 * NSPLIT), r = bid % (8 * NSPLIT);
    p = grp * 8 + (r & 7);
    half = r >> 3;
  }
  if (p >= P) return;
  const int ncol0 = half * NTILE;
  const int64_t r_begin = (int64_t)p * rows_per_wg;
  int64_t r_end = r_begin + rows_per_wg;
  if (r_end > M) r_end = M;
  const int nchunks = (int)((r_end - r_begin + KCH - 1) / KCH);

  // one staged chunk in registers: global -> registers -> LDS.  (Two stages, i.e. loads issued two chunk-times ahead
  // of their LDS store, were built and measured 2.4 % SLOWER on the 256x256 product: 0.563 vs 0.550 ms.)
  struct Stage {
    f32x4 xr[XV];
    f32x4 zr4[HEAD ? 1 : ZV];
    float zr1[HEAD ? ZV : 1];
  };
  // Chunk loads are buffer loads on the WORKGROUP'S OWN ROW RANGE: base = first row of the range (wave-uniform), chunk
  // offset = a scalar register, per-thread offset = one 32-bit register that never changes, and rows past the end of the
  // range read as zeros by the buffer bounds check -- no row predicates, no zero-selects and no 64-bit vector address
  // arithmetic in the loop (the predicated global_load form had 107 VALU instructions per 64 MFMAs, on the f32 lanes the
  // MFMAs run on).
  const int64_t range_rows = r_end - r_begin;
  const __amdgpu_buffer_rsrc_t rsX =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X + r_begin * KIN), 0, (int)(range_rows * KIN * 4), 0x00020000);
  constexpr int ZLD = DUAL ? kW : NOUT;             // row stride of a dZ array
  const __amdgpu_buffer_rsrc_t rsZ = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(HEAD ? dZ : dZ + r_begin * ZLD + (DUAL ? 0 : ncol0)), 0,
      HEAD ? 0 : (int)((range_rows * ZLD - (DUAL ? 0 : ncol0)) * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsZ2 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(DUAL ? dZ2 + r_begin * kW : dZ), 0, DUAL ? (int)(range_rows * kW * 4) : 0, 0x00020000);
  uint32_t xoff[XV], zoff[HEAD ? 1 : ZV];
#pragma unroll
  for (int i = 0; i < XV; ++i) {
    const int idx = tid + NT * i;
    xoff[i] = (uint32_t)((idx / (KIN / 4)) * KIN + (idx % (KIN / 4)) * 4) * 4u;
  }
  if (!HEAD) {
#pragma unroll
    for (int i = 0; i < ZV; ++i) {
      const int idx = tid + NT * i;
      const int row = idx / (NTILE / 4), c4 = idx % (NTILE / 4);
      zoff[i] = (DUAL ? (uint32_t)(row * kW + ((c4 * 4) & (kW - 1))) : (uint32_t)(row * NOUT + c4 * 4)) * 4u;
    }
  }
  auto load_chunk = [&](int ch, Stage& st) {
#pragma unroll
    for (int i = 0; i < XV; ++i)
      st.xr[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsX, xoff[i], ch * (KCH * KIN * 4), 0));
    if (!HEAD) {
      // DUAL: a thread's float4 column (tid + NT i) % 128 lies in the second array iff its wave is odd (NT = 256,
      // 128 float4 per row): a wave-uniform choice of the descriptor
      static_assert(!DUAL || (NT == 256 && NTILE == 2 * kW), "dual source: wave parity selects the array");
      const bool second = DUAL && (wave & 1);
#pragma unroll
      for (int i = 0; i < ZV; ++i)
        st.zr4[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(second ? rsZ2 : rsZ, zoff[i],
                                                                                     ch * (KCH * ZLD * 4), 0));
    } else {
      const int64_t r0 = r_begin + (int64_t)ch * KCH;
#pragma unroll
      for (int i = 0; i < ZV; ++i) {
        const int idx = tid + NT * i;
        const int row = idx / NTILE, col = idx % NTILE;
        const int64_t grow = r0 + row;
        float v = 0.f;
        if (grow < r_end) {
          if (col < C) v = dZ[grow * C + col];
          else if (col == C) v = d_raw_sigma[grow];
        }
        st.zr1[i] = v;
      }
    }
  };
  auto store_chunk = [&](int buf, const Stage& st) {
#pragma unroll
    for (int i = 0; i < XV; ++i) *reinterpret_cast<f32x4*>(&xs[buf][(tid + NT * i) * 4]) = st.xr[i];
    if (!HEAD) {
#pragma unroll
      for (int i = 0; i < ZV; ++i) *reinterpret_cast<f32x4*>(&zs[buf][(tid + NT * i) * 4]) = st.zr4[i];
    } else {
#pragma unroll
      for (int i = 0; i < ZV; ++i) zs[buf][tid + NT * i] = st.zr1[i];
    }
  };

  f32x16 acc[RB][CB];
#pragma unroll
  for (int r = 0; r < RB; ++r)
#pragma unroll
    for (int c = 0; c < CB; ++c)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[r][c][i] = 0.f;

  // computes chunk `ch` out of LDS buffer `buf`; on the way: loads chunk `ld_ch` into `ld` (if do_ld) and stores the
  // chunk held in `stg` into the other LDS buffer (if do_st)
  auto run_chunk = [&](int ch, int buf, Stage& ld, int ld_ch, bool do_ld, const Stage& stg, bool do_st) {
    (void)ch;
    if (SCHED == 0 && do_ld) load_chunk(ld_ch, ld);
    const float* xa = &xs[buf][(lane >> 5) * KIN + (wr * RB) * 32 + (lane & 31)];
    const float* zb = &zs[buf][(lane >> 5) * NTILE + (wc * CB) * 32 + (lane & 31)];
    // (A chunk staged as [32-column block][row][32 columns], which turns every operand read into ds_read2st64_b32 immediates
    // off ONE base register -- 10 instead of 18 VALU per 64 MFMAs -- measured 0.65 % slower per step, round 3: not kept.
    // Reading through volatile address_space(3) pointers -- every read a ds_read_b32 with a 16-bit immediate, NO vector ALU
    // between the 64 MFMAs of a chunk, but 48 LDS instructions instead of 24 -- measured 0.8 % slower for this kernel
    // (3.537 -> 3.565 ms): with two workgroups per CU the other workgroup's MFMAs fill the slots the v_adds cost.)
    // operands of k-step s+1 are read from LDS before the MFMAs of k-step s (order pinned: hipcc
    // otherwise sinks every ds_read to just before its use and waits lgkmcnt(0) every 4 MFMAs)
    float a0[RB], b0[CB], a1[RB], b1[CB];
    auto read_step = [&](int kk, float (&a)[RB], float (&b)[CB]) {
#pragma unroll
      for (int r = 0; r < RB; ++r) a[r] = xa[kk * KIN + r * 32];
#pragma unroll
      for (int c = 0; c < CB; ++c) b[c] = zb[kk * NTILE + c * 32];
    };
    auto mfma_step = [&](const float (&a)[RB], const float (&b)[CB]) {
#pragma unroll
      for (int r = 0; r < RB; ++r)
#pragma unroll
        for (int c = 0; c < CB; ++c)
          acc[r][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r], b[c], acc[r][c], 0, 0, 0);
    };
    read_step(0, a0, b0);
#pragma unroll
    for (int kk = 0; kk < KCH; kk += 4) {
      read_step(kk + 2, a1, b1);
      __builtin_amdgcn_sched_barrier(0);
      mfma_step(a0, b0);
      __builtin_amdgcn_sched_barrier(0);
      // SCHED 1 (the skinny, HBM-heavy products; the 256x256 product measured 2 % slower with it): the loads
      // (address arithmetic + 6 loads) are issued under the first MFMA group of this chunk, the LDS stores under the
      // last one: the stretch between a chunk's last MFMA and the next chunk's first is then only "barrier + first
      // operand reads"
      if (SCHED != 0 && kk == 0 && do_ld) load_chunk(ld_ch, ld);
      if (SCHED != 0 && kk == 0) __builtin_amdgcn_sched_barrier(0);
      read_step(kk + 4 < KCH ? kk + 4 : kk + 2, a0, b0);   // harmless re-read on the last trip
      __builtin_amdgcn_sched_barrier(0);
      if (SCHED != 0 && kk + 4 >= KCH && do_st) store_chunk(buf ^ 1, stg);
      if (SCHED != 0 && kk + 4 >= KCH) __builtin_amdgcn_sched_barrier(0);
      mfma_step(a1, b1);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (SCHED == 0 && do_st) store_chunk(buf ^ 1, stg);
    __syncthreads();
  };

  // Zero-row skipping (SPARSE instantiation, chunk_live != NULL): the backward(data) kernel flagged every 16-row chunk that
  // has a row with a non-zero upstream gradient; all other rows have dz = 0 in every layer (and may not even have been
  // written), so their chunks add exactly nothing.  Wave 0 compacts the range's live chunks, in order, into an LDS list; the
  // loop then walks the list instead of 0..nchunks-1 (same accumulation order over the chunks that contribute: bit-identical
  // sums).  A separate instantiation: in the dense kernel the chunk index stays the loop counter (a scalar; as a value read
  // from LDS it cost the dense 256x256 product 5 %, measured).
  // (ranges longer than kMaxLiveChunks chunks never reach a SPARSE launch: wgrad_skip_supported / the launcher's check --
  // in skipping mode backward(data) leaves the dz rows of dead tiles unwritten, so a dense walk here would read garbage)
  constexpr int kMaxLive = SPARSE ? kMaxLiveChunks : 1;
  __shared__ uint16_t live_list[kMaxLive];
  __shared__ int live_count;
  const bool sparse = SPARSE && chunk_live != nullptr && nchunks <= kMaxLive;
  int n_run = nchunks;
  if (SPARSE && sparse) {
    if (wave == 0) {
      const uint8_t* fl = chunk_live + r_begin / KCH;
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
    for (int i = 0; i < n_run; ++i) {
      const bool more = i + 1 < n_run;
      run_chunk(i, i & 1, st, more ? chunk_at(i + 1) : 0, more, st, more);
    }
  }

  float* out = slab + (int64_t)p * KIN * NOUT;
#pragma unroll
  for (int r = 0; r < RB; ++r)
#pragma unroll
    for (int c = 0; c < CB; ++c) {
      const int n = ncol0 + (wc * CB + c) * 32 + (lane & 31);
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int i = (wr * RB + r) * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
        out[(int64_t)i * NOUT + n] = acc[r][c][reg];
      }
    }
}

// One launch for every slab reduction of a weight-gradient pass: blockIdx.y selects the job.
// A job sums P slabs of kin x nout and scatters column ranges A and B of the first rows_valid rows to two leaves.
struct ReduceJob {
  const float* slab;
  int P, kin, nout, rows_valid;
  float* dst_a; int col0_a, ncols_a, ld_a;
  float* dst_b; int col0_b, ncols_b, ld_b;      // dst_b == nullptr: no second range
};
struct ReduceJobs {
  ReduceJob job[kDepth + 1];                    // Dense_1..7, enc-based pair (Dense_0 | Dense_5 skip rows), heads
  const float* dbias_partial;                   // + the bias gradients as job kDepth + 1: [slot][9][256] tile partials
  const uint8_t* tile_live;                     //   one byte per slot (0: the tile was skipped, its partial was not written)
  int64_t dbias_tiles;
  const float* dz_colsum;                       //   bf16x6 with kBiasFromWgrad: Dense_1..7 from [layer - 1][range][256] instead
  int colsum_ranges;
  int deg;
  float* grads;
};

__global__ __launch_bounds__(256) void reduce_jobs_kernel(ReduceJobs J) {
  __shared__ f32x4 red[4][64];
  if (blockIdx.y == kDepth + 1) {
    // bias gradients: block = (layer 0..8, 8-column group), 32 sub-threads per column stride over the tile slots with 8
    // independent loads in flight (the slots are thousands at 4096 rays: a serial loop over them is pure latency), then a
    // fixed-order LDS sum.  Dead slots (tile_live == 0: skipped tiles, whose partial was never written) are left out; a
    // live tile's partial of a dead ROW range is an exact zero, so dense and skipping passes give the same bits.
    if (blockIdx.x >= 9 * 32) return;
    float (*redb)[8] = reinterpret_cast<float (*)[8]>(&red[0][0]);
    const int l = blockIdx.x / 32, cg = blockIdx.x % 32;
    const int c = threadIdx.x & 7, tsub = threadIdx.x >> 3;
    const int col = cg * 8 + c;
    const int64_t n = J.dbias_tiles;
    float s = 0.f;
    if (J.dz_colsum && l >= 1 && l < kDepth) {
      // the weight-gradient kernel's column sums of dz_l, one per row range: the same 32-way strided fixed-order sum
      const float* __restrict__ src = J.dz_colsum + (int64_t)(l - 1) * J.colsum_ranges * kW + col;
      for (int pp = tsub; pp < J.colsum_ranges; pp += 32) s += src[(int64_t)pp * kW];
    } else
    for (int64_t t0 = tsub; t0 < n; t0 += 32 * 8) {
      bool lv[8];
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int64_t t = t0 + 32 * k;
        lv[k] = t < n && J.tile_live[t] != 0;
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int64_t t = t0 + 32 * k;
        v[k] = lv[k] ? J.dbias_partial[(t * 9 + l) * kW + col] : 0.f;
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) s += v[k];
    }
    redb[tsub][c] = s;
    __syncthreads();
    if (tsub == 0) {
      float tot = 0.f;
#pragma unroll
      for (int i = 0; i < 32; ++i) tot += redb[i][c];
      if (l < 8) {
        J.grads[leaf_bias_off(l, J.deg) + col] = tot;
      } else {
        const int C = rgb_channels(J.deg);
        if (col < C) J.grads[leaf_bias_off(9, J.deg) + col] = tot;
        else if (col == C) J.grads[leaf_bias_off(8, J.deg)] = tot;
      }
    }
    return;
  }
  const ReduceJob& j = J.job[blockIdx.y];
  const int v = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int64_t e4 = (int64_t)blockIdx.x * 64 + v;
  const int64_t stride4 = (int64_t)j.kin * j.nout / 4;
  if ((int64_t)blockIdx.x * 64 >= stride4) return;            // whole block past this job's slab
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (e4 < stride4) {
    const f32x4* __restrict__ src = reinterpret_cast<const f32x4*>(j.slab) + e4;
#pragma unroll 8
    for (int p = q; p < j.P; p += 4) s += src[p * stride4];
  }
  red[q][v] = s;
  __syncthreads();
  if (q == 0 && e4 < stride4) {
    f32x4 t = red[0][v];
    t += red[1][v]; t += red[2][v]; t += red[3][v];
    const int64_t e = e4 * 4;
    const int i = (int)(e / j.nout), n0 = (int)(e % j.nout);
    if (i < j.rows_valid) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int n = n0 + k;
        if (n >= j.col0_a && n < j.col0_a + j.ncols_a) j.dst_a[(int64_t)i * j.ld_a + (n - j.col0_a)] = t[k];
        else if (j.dst_b && n >= j.col0_b && n < j.col0_b + j.ncols_b) j.dst_b[(int64_t)i * j.ld_b + (n - j.col0_b)] = t[k];
      }
    }
  }
}

// live chunks of a pass (reporting only: pxo_train_backward_work)
__global__ void count_live_kernel(const uint8_t* __restrict__ fl, int64_t n, unsigned long long* __restrict__ out) {
  unsigned long long c = 0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) c += fl[i] != 0;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, c);
}
int launch_count_live(const uint8_t* chunk_live, int64_t n, unsigned long long* out, hipStream_t s) {
  if (n == 0) return PXO_OK;
  hipLaunchKernelGGL(count_live_kernel, dim3(64), dim3(256), 0, s, chunk_live, n, out);
  return check_launch("count_live");
}


static void split_rows(int64_t M, int64_t target, int64_t* rows_per_wg, int* P) {
  int64_t rpw = (M + target - 1) / target;
  rpw = (rpw + kKC - 1) / kKC * kKC;
  if (rpw < kKC) rpw = kKC;
  *rows_per_wg = rpw;
  *P = (int)((M + rpw - 1) / rpw);
}

// The split of a pass of M rows: row ranges per layer of the 256x256 products / of the skinny products.
//  * big passes (>= 1024 rows per CU): one range per CU and layer (NL whole waves of workgroups on the 2 num_cus() slots);
//  * smaller ones: 4 num_cus() / NL ranges (4 waves) -- round 3, rays/s at 4096 / 1024 / 512 rays per step on 256 CUs:
//    256 ranges 161.0 k / 141.6 k / 123.6 k, 146 ranges 156.5 k / 144.3 k / 127.9 k (one launch per layer: 159.7 k /
//    139.9 k / 120.3 k): long ranges lose on big passes (the two column halves drift apart and the shared operand stops
//    hitting L2), short ones pay 64 MB of slab traffic per layer;
//  * the skinny products run two workgroups per CU: 2 num_cus() ranges, num_cus() below 512 rows per CU (round 3, ms per step
//    at 512 / 1024 / 4096 rays: 512 ranges 3.732 / 6.686 / 24.19, 256 ranges 3.716 / 6.689 / 24.28, 384: 3.739 / 6.722 /
//    24.41, 128: slower).
// pxo_set_tuning(PXO_TUNE_WGRAD_RANGES / PXO_TUNE_WGRAD_SKINNY_RANGES) overrides either (A/B sessions).
struct WgradSplit { int64_t rpw_main; int P_main; int64_t rpw_skinny; int P_skinny; };
static WgradSplit wgrad_split(int64_t M) {
  constexpr int NL = kDepth - 1;
  WgradSplit w;
  int ranges = M >= (int64_t)1024 * num_cus() ? num_cus() : (4 * num_cus() / NL > 0 ? 4 * num_cus() / NL : 1);
  if (tune_wgrad_ranges() > 0) ranges = tune_wgrad_ranges();
  if (ranges > num_cus()) ranges = num_cus();
  split_rows(M, ranges, &w.rpw_main, &w.P_main);
  int skinny = M < (int64_t)512 * num_cus() ? num_cus() : 2 * num_cus();
  if (tune_wgrad_skinny_ranges() > 0) skinny = tune_wgrad_skinny_ranges();
  if (skinny > 2 * num_cus()) skinny = 2 * num_cus();
  split_rows(M, skinny, &w.rpw_skinny, &w.P_skinny);
  return w;
}
// zero-row skipping needs every row range to fit the workgroup's live-chunk list
bool wgrad_skip_supported(int64_t M) {
  if (M <= 0) return true;
  const WgradSplit w = wgrad_split(M);
  return w.rpw_main <= (int64_t)kMaxLiveChunks * kLiveRows && w.rpw_skinny <= (int64_t)kMaxLiveChunks * kLiveRows;
}

size_t wgrad_workspace_bytes(const PxoCfg* cfg, int64_t M) {
  // one KIN x NOUT slab per workgroup: 256 x 256 for up to num_cus() row ranges, or 64 x 512 / 256 x 96 for up to
  // 2 num_cus() (the skinny products run two workgroups per CU); the size does not depend on M (two passes of
  // different M share one workspace)
  (void)cfg; (void)M;
  // Dense_1..7 in one launch: a slab set per layer; then the enc-based pair's and the heads' slabs (all reduced together)
  // + the bf16x6 weight-gradient kernel's column sums of dz_1..7: [7][ranges][256]
  return ((size_t)(kDepth - 1) * num_cus() * kW * kW + (size_t)2 * num_cus() * kEncPad * 2 * kW +
          (size_t)2 * num_cus() * kW * 32 * 3 + (size_t)(kDepth - 1) * num_cus() * kW) * sizeof(float);
}

template <int NHB>
static void launch_head_wgrad(const float* X, const float* d_raw_rgb, const float* d_raw_sigma, int C,
                              int64_t M, int64_t rpw, int P, float* slab, const uint8_t* live, hipStream_t s) {
  // 4 waves, each 64 rows x all head columns (4 LDS operand reads per 4 MFMAs instead of 3 per 2), 40 KB of LDS:
  // several workgroups per CU
  if (live)
    hipLaunchKernelGGL((wgrad_kernel<kW, 32 * NHB, 4, 1, true, 256, 16, 1, false, 1, true>), dim3(P), dim3(256), 0, s, X, d_raw_rgb,
                       d_raw_sigma, C, M, rpw, P, slab, (const float*)nullptr, 1, (int64_t)0, live);
  else
    hipLaunchKernelGGL((wgrad_kernel<kW, 32 * NHB, 4, 1, true, 256, 16, 1, false, 1>), dim3(P), dim3(256), 0, s, X, d_raw_rgb,
                       d_raw_sigma, C, M, rpw, P, slab);
}

int launch_mlp_bwd_weights(const PxoCfg* cfg, const float* acts, const float* enc, const float* dz,
                           const float* d_raw_rgb, const float* d_raw_sigma,
                           const float* dbias_partial, int64_t M, float* grads, void* ws,
                           size_t ws_bytes, const uint8_t* chunk_live, hipStream_t s, int flags) {
  if (M == 0) return PXO_OK;
  const int deg = cfg->sh_deg;
  const int C = rgb_channels(deg);
  const int nhb = head_blocks(deg);
  const int64_t MW = M * kW;
  constexpr int NL = kDepth - 1;
  (void)NL;
  if (ws_bytes < wgrad_workspace_bytes(cfg, M)) {
    set_error("wgrad workspace too small: %zu < %zu", ws_bytes, wgrad_workspace_bytes(cfg, M));
    return PXO_ERR_WORKSPACE;
  }
  // the skinny products (enc-based pair, heads): two workgroups per CU.  Issuing them on side streams beside the 256x256 launch was measured too (round 3,
  // 512 rays per step): 4.05 vs 3.96 ms per step -- the HBM-leaning workgroups take CU slots from the MFMA-bound ones
  // early and the step gets longer, so the three launches stay in stream order.
  const WgradSplit split = wgrad_split(M);
  if (chunk_live && !wgrad_skip_supported(M)) {
    set_error("mlp_bwd_weights: zero-row skipping with row ranges of %lld / %lld rows (> %d chunks of %d rows per workgroup); "
              "the caller must run this pass dense (wgrad_skip_supported)", (long long)split.rpw_main,
              (long long)split.rpw_skinny, kMaxLiveChunks, kLiveRows);
    return PXO_ERR_UNSUPPORTED;
  }
  const int64_t rpw2 = split.rpw_skinny; const int P2 = split.P_skinny;
  const float* h7 = acts + (int64_t)7 * MW;
  auto head = [&](float* slab) {
    KernelTimer timer(PXO_PROF_WGRAD_OTHER, M, s);
    if (nhb == 1) launch_head_wgrad<1>(h7, d_raw_rgb, d_raw_sigma, C, M, rpw2, P2, slab, chunk_live, s);
    else if (nhb == 2) launch_head_wgrad<2>(h7, d_raw_rgb, d_raw_sigma, C, M, rpw2, P2, slab, chunk_live, s);
    else launch_head_wgrad<3>(h7, d_raw_rgb, d_raw_sigma, C, M, rpw2, P2, slab, chunk_live, s);
  };
  // Dense_0 and the skip rows 256..318 of Dense_5 in one pass over enc: enc^T [dz_0 | dz_5]  (63 valid input rows)
  auto enc_pair = [&](float* slab) {
    KernelTimer timer(PXO_PROF_WGRAD_OTHER, M, s);
    if (chunk_live)
      hipLaunchKernelGGL((wgrad_kernel<kEncPad, 2 * kW, 1, 4, false, 256, 16, 1, true, 1, true>), dim3(P2), dim3(256), 0, s, enc, dz,
                         nullptr, 0, M, rpw2, P2, slab, dz + (int64_t)5 * MW, 1, (int64_t)0, chunk_live);
    else
      hipLaunchKernelGGL((wgrad_kernel<kEncPad, 2 * kW, 1, 4, false, 256, 16, 1, true, 1>), dim3(P2), dim3(256), 0, s, enc, dz,
                         nullptr, 0, M, rpw2, P2, slab, dz + (int64_t)5 * MW);
  };
  float* const g0 = grads + leaf_kernel_off(0, deg);
  float* const g5skip = grads + leaf_kernel_off(5, deg) + (int64_t)kW * kW;
  float* const g8 = grads + leaf_kernel_off(8, deg);
  float* const g9 = grads + leaf_kernel_off(9, deg);
  // Dense_1..7: h_{l-1}^T dz_l (for l = 5 these are the first 256 input rows) in ONE launch: acts and dz are [8][M,256]
  // stacks, so layer l is "group l-1" of the kernel with layer_stride = M*256; then ONE launch reduces every slab set of
  // the pass (and the bias partials).  14 + 12 launches per step fewer than a launch per product, and the layers'
  // ramp-ups and tails overlap.
  // Row ranges per layer: wgrad_split above (the grid is NL x ranges x 2 column halves on 2 num_cus() slots, all of equal work).
  const int64_t rpwb = split.rpw_main; const int Pb = split.P_main;
  float* const slab_main = reinterpret_cast<float*>(ws);
  float* const slab_enc = slab_main + (size_t)NL * num_cus() * kW * kW;
  float* const slab_head = slab_enc + (size_t)2 * num_cus() * kEncPad * 2 * kW;
  float* const dz_colsum = slab_head + (size_t)2 * num_cus() * kW * 32 * 3;
  const bool x6_main = cfg->mlp_precision == PXO_MLP_BF16X6 && tune_x6_wgrad() != 0;
  if ((flags & kBiasFromWgrad) && !x6_main) {
    set_error("mlp_bwd_weights: kBiasFromWgrad without the bf16x6 weight-gradient kernel");
    return PXO_ERR_ARG;
  }
  enc_pair(slab_enc);
  {
    KernelTimer timer(PXO_PROF_WGRAD_MAIN, M * NL, s);
    if (x6_main)
      launch_wgrad_main_x6(acts, dz + MW, M, rpwb, Pb, slab_main, NL, MW, chunk_live, dz_colsum, s);
    else if (chunk_live)
      hipLaunchKernelGGL((wgrad_kernel<kW, kW, 2, 2, false, 256, 16, 2, false, 0, true>), dim3(NL * ((Pb + 7) / 8) * 16), dim3(256),
                         0, s, acts, dz + MW, nullptr, 0, M, rpwb, Pb, slab_main, nullptr, NL, MW, chunk_live);
    else
      hipLaunchKernelGGL((wgrad_kernel<kW, kW, 2, 2, false, 256, 16, 2>), dim3(NL * ((Pb + 7) / 8) * 16), dim3(256), 0, s,
                         acts, dz + MW, nullptr, 0, M, rpwb, Pb, slab_main, nullptr, NL, MW);
  }
  head(slab_head);
  ReduceJobs J;
  for (int l = 1; l < kDepth; ++l)
    J.job[l - 1] = ReduceJob{slab_main + (size_t)(l - 1) * Pb * kW * kW, Pb, kW, kW, kW,
                             grads + leaf_kernel_off(l, deg), 0, kW, kW, nullptr, 0, 0, 0};
  J.job[NL] = ReduceJob{slab_enc, P2, kEncPad, 2 * kW, kEnc, g0, 0, kW, kW, g5skip, kW, kW, kW};
  J.job[NL + 1] = ReduceJob{slab_head, P2, kW, 32 * nhb, kW, g9, 0, C, C, g8, C, 1, 1};
  J.dbias_partial = dbias_partial;
  J.tile_live = dbias_tile_live(dbias_partial, M);
  J.dbias_tiles = (int64_t)mlp_bwd_partials(M);
  J.dz_colsum = (flags & kBiasFromWgrad) ? dz_colsum : nullptr;
  J.colsum_ranges = Pb;
  J.deg = deg;
  J.grads = grads;
  // (Issuing the coarse pass's reduction on a lowest-priority side stream, to run in the tail of the fine pass's backward
  // kernel, was measured in round 3: 3.896 vs 3.904 ms per step at 512 rays, 25.26 vs 25.23 at 4096 -- nothing; not kept.)
  hipLaunchKernelGGL(reduce_jobs_kernel, dim3(9 * 32 > kW * kW / 4 / 64 ? 9 * 32 : kW * kW / 4 / 64, kDepth + 2), dim3(256), 0, s, J);
  return check_launch("mlp_bwd_weights");
}


}  // namespace pxo
