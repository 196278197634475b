// Ceiling probe 3: geometry and B-path variants of the fused-MLP GEMM loop (no epilogue).
//  G0: 64x64 per wave (RB=2,CB=2), 4 waves, 2 WG/CU, B global->VGPR      (= current kernel)
//  G1: 128x32 per wave (RB=4,CB=1), 8 waves, 1 WG/CU, B global->VGPR
//  G2: G0 with B via LDS-DMA (global_load_lds 16 B/lane into a per-wave ring) + ds_read_b128
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kLDA = 260;
#define PIN() __builtin_amdgcn_sched_barrier(0)

template <int RB, int CB, int NW, int ROWS>
__global__ __launch_bounds__(NW * 64, 2) void gemm_regs(const float* __restrict__ wimg, const float* __restrict__ in,
                                                       float* __restrict__ out, int layers) {
  __shared__ __attribute__((aligned(16))) float lds[ROWS * kLDA];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < ROWS * kLDA; i += NW * 64) lds[i] = in[i % 4096];
  __syncthreads();
  const float* arow = lds + (lane & 31) * kLDA + (lane >> 5) * 4;
  f32x16 acc[RB][CB];
  for (int r = 0; r < RB; ++r) for (int c = 0; c < CB; ++c) for (int e = 0; e < 16; ++e) acc[r][c][e] = 0.f;
  auto mf = [&](const f32x4 (&a)[RB], const f32x4 (&b)[CB]) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < RB; ++r)
#pragma unroll
        for (int c = 0; c < CB; ++c) acc[r][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r][j], b[c][j], acc[r][c], 0, 0, 0);
  };
  for (int l = 0; l < layers; ++l) {
    const f32x4* wp = reinterpret_cast<const f32x4*>(wimg) + (size_t)(l & 7) * 32 * 8 * 64 + (wave * CB) * 64 + lane;
    f32x4 a0[RB], a1[RB], b0[CB], b1[CB], b2[CB], b3[CB];
    auto la = [&](int g, f32x4 (&a)[RB]) {
#pragma unroll
      for (int r = 0; r < RB; ++r) a[r] = *(const f32x4*)(arow + r * 32 * kLDA + g * 8);
    };
    auto lb = [&](int g, f32x4 (&b)[CB]) {
#pragma unroll
      for (int c = 0; c < CB; ++c) b[c] = wp[(size_t)g * 512 + c * 64];
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
  for (int r = 0; r < RB; ++r) for (int c = 0; c < CB; ++c) for (int e = 0; e < 16; ++e) s += acc[r][c][e];
  out[blockIdx.x * NW * 64 + tid] = s;
}

// B via LDS-DMA: per wave a ring of RING slots x (2 col-blocks x 1 KiB); slot s holds k-group g = s mod RING
template <int RING>
__global__ __launch_bounds__(256, 2) void gemm_dma(const float* __restrict__ wimg, const float* __restrict__ in,
                                                  float* __restrict__ out, int layers) {
  __shared__ __attribute__((aligned(16))) float lds[64 * 256 + 4 * RING * 512];   // swizzle-free A (stride 256) for the probe
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 64 * 256; i += 256) lds[i] = in[i % 4096];
  __syncthreads();
  float* ring = lds + 64 * 256 + wave * RING * 512;            // this wave's private ring (floats)
  const float* arow = lds + (lane & 31) * 256 + (lane >> 5) * 4;  // (bank conflicts ignored in this probe)
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
    const float* wl = wimg + (size_t)(l & 7) * 65536 + (wave * 2) * 256;   // floats; k-group stride 2048 floats
    auto dma = [&](int g) {   // two 1 KiB pieces (col-blocks) of k-group g into ring slot g % RING
      float* dst = ring + (g % RING) * 512;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wl + (size_t)g * 2048 + lane * 4),
                                       (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wl + (size_t)g * 2048 + 256 + lane * 4),
                                       (__attribute__((address_space(3))) void*)(dst + 256), 16, 0, 0);
    };
    f32x4 a0[2], a1[2], b0[2], b1[2];
    auto la = [&](int g, f32x4 (&a)[2]) { a[0] = *(const f32x4*)(arow + g * 8); a[1] = *(const f32x4*)(arow + 32 * 256 + g * 8); };
    auto lb = [&](int g, f32x4 (&b)[2]) {
      const float* src = ring + (g % RING) * 512 + lane * 4;
      b[0] = *(const f32x4*)(src); b[1] = *(const f32x4*)(src + 256);
    };
    // prologue: RING-1 groups in flight
    for (int g = 0; g < RING - 1; ++g) dma(g);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (RING - 2)) : "memory");
    la(0, a0); lb(0, b0);
    for (int g = 0; g < 32; g += 2) {
      dma(g + RING - 1 < 32 ? g + RING - 1 : 31);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (RING - 2)) : "memory");   // group g+1 landed
      la(g + 1, a1); lb(g + 1, b1); PIN(); mf(a0, b0); PIN();
      dma(g + RING < 32 ? g + RING : 31);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (RING - 2)) : "memory");   // group g+2 landed
      la(g + 2 < 32 ? g + 2 : 31, a0); lb(g + 2 < 32 ? g + 2 : 31, b0); PIN(); mf(a1, b1); PIN();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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
  hipMalloc(&w, 8 * 65536 * 4); hipMalloc(&in, 8192 * 4); hipMalloc(&out, 1024 * 512 * 4);
  hipMemcpy(w, h.data(), 8 * 65536 * 4, hipMemcpyHostToDevice);
  hipMemcpy(in, h.data() + 8 * 65536, 8192 * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int v = 0; v < 5; ++v) {
    float ms = 0; int blocks = (v == 1 || v == 4) ? 256 : 512; int rows = (v == 1 || v == 4) ? 128 : 64;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      if (v == 0) hipLaunchKernelGGL((gemm_regs<2, 2, 4, 64>), dim3(512), dim3(256), 0, 0, w, in, out, layers);
      if (v == 1) hipLaunchKernelGGL((gemm_regs<4, 1, 8, 128>), dim3(256), dim3(512), 0, 0, w, in, out, layers);
      if (v == 2) hipLaunchKernelGGL((gemm_dma<4>), dim3(512), dim3(256), 0, 0, w, in, out, layers);
      if (v == 3) hipLaunchKernelGGL((gemm_dma<3>), dim3(512), dim3(256), 0, 0, w, in, out, layers);
      if (v == 4) hipLaunchKernelGGL((gemm_regs<4, 1, 4, 128>), dim3(256), dim3(256), 0, 0, w, in, out, layers);   // ONE wave per SIMD
      hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
    }
    double flop = (double)blocks * rows * 256.0 * (v == 4 ? 128.0 : 256.0) * 2.0 * layers;
    const char* names[] = {"G0 64x64/wave 4w 2WG/CU B->VGPR", "G1 128x32/wave 8w 1WG/CU B->VGPR", "G2 G0 + B via LDS-DMA ring4", "G3 G0 + B via LDS-DMA ring3", "G4 128x32/wave 4w (one wave per SIMD) 1WG/CU B->VGPR"};
    printf("%s: %.3f ms  %.1f TFLOP/s (%.1f%%)  err=%s\n", names[v], ms, flop / ms / 1e9, 100 * flop / ms / 1e9 / 157.3, hipGetErrorString(hipGetLastError()));
  }
  return 0;
}
