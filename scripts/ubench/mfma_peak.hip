// Standalone ceiling probe: v_mfma_f32_32x32x2_f32 issue rate under sustained load on MI355X.
// variants: waves per SIMD (1|2), accumulators per wave (4), with/without LDS reads feeding A.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool LDSFEED>
__global__ __launch_bounds__(256, 2) void mfma_loop(const float* __restrict__ in, float* __restrict__ out, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[64 * 260];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 64 * 260; i += 256) lds[i] = in[i % 4096];
  __syncthreads();
  f32x16 acc[4];
  for (int a = 0; a < 4; ++a) for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
  f32x4 a0 = *(const f32x4*)(in + lane * 4), a1 = *(const f32x4*)(in + 256 + lane * 4);
  f32x4 b0 = *(const f32x4*)(in + 512 + lane * 4), b1 = *(const f32x4*)(in + 768 + lane * 4);
  const float* arow = lds + (lane & 31) * 260 + (lane >> 5) * 4;
  for (int it = 0; it < iters; ++it) {
    if (LDSFEED) {
      a0 = *(const f32x4*)(arow + (it & 31) * 8);
      a1 = *(const f32x4*)(arow + 32 * 260 + (it & 31) * 8);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b0[j], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b1[j], acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b0[j], acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b1[j], acc[3], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  float s = 0.f;
  for (int a = 0; a < 4; ++a) for (int e = 0; e < 16; ++e) s += acc[a][e];
  out[blockIdx.x * 256 + tid] = s;
}

int main() {
  const int iters = 20000;
  std::vector<float> h(8192);
  for (auto& x : h) x = (float)rand() / RAND_MAX * 2.f - 1.f;
  float *in, *out;
  hipMalloc(&in, 8192 * 4); hipMalloc(&out, 1024 * 256 * 4);
  hipMemcpy(in, h.data(), 8192 * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int lds = 0; lds < 2; ++lds)
    for (int blocks : {256, 512}) {
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        if (lds) hipLaunchKernelGGL(mfma_loop<true>, dim3(blocks), dim3(256), 0, 0, in, out, iters);
        else hipLaunchKernelGGL(mfma_loop<false>, dim3(blocks), dim3(256), 0, 0, in, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double flop = (double)blocks * 4 * iters * 16 * 2.0 * 32 * 32 * 2;
        if (rep == 2) printf("ldsfeed=%d blocks=%d (%d wave/SIMD): %.3f ms  %.1f TFLOP/s  (%.1f%% of 157.3)\n", lds, blocks,
                             blocks / 256, ms, flop / ms / 1e9, 100 * flop / ms / 1e9 / 157.3);
      }
    }
  return 0;
}
