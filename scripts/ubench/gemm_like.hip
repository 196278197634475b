// Ceiling probe 2: the fused-MLP GEMM loop in isolation.  A from LDS (64x256 tile), B streamed from a
// 2 MB fragment-ordered image (L2-resident), 64x64 per wave, 2 workgroups per CU.  Variants: B from
// global vs B constant in registers; prefetch pinned.  No epilogue, no stores.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kLDA = 260;
#define PIN() __builtin_amdgcn_sched_barrier(0)

template <int MODE>  // 0: B from global (2-ahead), 1: B fixed registers, 2: B from global, no A reads (A fixed)
__global__ __launch_bounds__(256, 2) void gemm_like(const float* __restrict__ wimg, const float* __restrict__ in,
                                                   float* __restrict__ out, int layers) {
  __shared__ __attribute__((aligned(16))) float lds[64 * kLDA];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 64 * kLDA; i += 256) lds[i] = in[i % 4096];
  __syncthreads();
  const float* arow = lds + (lane & 31) * kLDA + (lane >> 5) * 4;
  f32x16 acc[2][2];
  for (int r = 0; r < 2; ++r) for (int c = 0; c < 2; ++c) for (int e = 0; e < 16; ++e) acc[r][c][e] = 0.f;
  auto mf = [&](const f32x4 (&a)[2], const f32x4 (&b)[2]) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < 2; ++c) acc[r][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r][j], b[c][j], acc[r][c], 0, 0, 0);
  };
  for (int l = 0; l < layers; ++l) {
    const f32x4* wp = reinterpret_cast<const f32x4*>(wimg) + (size_t)(l & 7) * 32 * 8 * 64 + (wave * 2) * 64 + lane;
    f32x4 a0[2], a1[2], b0[2], b1[2], b2[2], b3[2];
    auto la = [&](int g, f32x4 (&a)[2]) {
      if (MODE == 2) { a[0] = *(const f32x4*)(arow); a[1] = *(const f32x4*)(arow + 32 * kLDA); return; }
      a[0] = *(const f32x4*)(arow + g * 8); a[1] = *(const f32x4*)(arow + 32 * kLDA + g * 8);
    };
    auto lb = [&](int g, f32x4 (&b)[2]) {
      if (MODE == 1) { b[0] = wp[0]; b[1] = wp[64]; return; }
      b[0] = wp[(size_t)g * 512]; b[1] = wp[(size_t)g * 512 + 64];
    };
    lb(0, b0); lb(1, b1); la(0, a0);
    for (int g = 0; g < 32; g += 4) {
      la(g + 1, a1); lb(g + 2, b2); PIN(); mf(a0, b0); PIN();
      la(g + 2, a0); lb(g + 3, b3); PIN(); mf(a1, b1); PIN();
      la(g + 3, a1); lb(g + 4 < 31 ? g + 4 : 31, b0); PIN(); mf(a0, b2); PIN();
      la(g + 4 < 31 ? g + 4 : 31, a0); lb(g + 5 < 31 ? g + 5 : 31, b1); PIN(); mf(a1, b3); PIN();
    }
  }
  float s = 0.f;
  for (int r = 0; r < 2; ++r) for (int c = 0; c < 2; ++c) for (int e = 0; e < 16; ++e) s += acc[r][c][e];
  out[blockIdx.x * 256 + tid] = s;
}

int main() {
  const int layers = 400;
  std::vector<float> h(8 * 65536 + 8192);
  for (auto& x : h) x = (float)rand() / RAND_MAX * 2.f - 1.f;
  float *w, *in, *out;
  hipMalloc(&w, 8 * 65536 * 4); hipMalloc(&in, 8192 * 4); hipMalloc(&out, 1024 * 256 * 4);
  hipMemcpy(w, h.data(), 8 * 65536 * 4, hipMemcpyHostToDevice);
  hipMemcpy(in, h.data() + 8 * 65536, 8192 * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 3; ++mode)
    for (int blocks : {256, 512}) {
      float ms = 0;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(gemm_like<0>, dim3(blocks), dim3(256), 0, 0, w, in, out, layers);
        if (mode == 1) hipLaunchKernelGGL(gemm_like<1>, dim3(blocks), dim3(256), 0, 0, w, in, out, layers);
        if (mode == 2) hipLaunchKernelGGL(gemm_like<2>, dim3(blocks), dim3(256), 0, 0, w, in, out, layers);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
      }
      double flop = (double)blocks * 4 * layers * 32 * 16 * 2.0 * 32 * 32 * 2;
      printf("mode=%d (%s) blocks=%d: %.3f ms  %.1f TFLOP/s (%.1f%%)\n", mode,
             mode == 0 ? "A lds, B global" : mode == 1 ? "A lds, B regs" : "A fixed, B global", blocks, ms,
             flop / ms / 1e9, 100 * flop / ms / 1e9 / 157.3);
    }
  return 0;
}
