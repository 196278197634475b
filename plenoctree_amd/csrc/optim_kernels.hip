// Loss, statistics and Adam for the NeRF-SH train step on gfx950.
// Reference: nerf_sh/train.py:77-114 (loss terms), nerf_sh/nerf/utils.py:384-393 (psnr),
// flax.optim.Adam (third-party, flax>=0.3.1; call sites nerf_sh/nerf/models.py:44, train.py:119).
#include "pxo_common.h"

namespace pxo {

constexpr int kRedThreads = 1024;

// fixed-order block reduction (deterministic)
__device__ __forceinline__ float block_sum(float v, float* red) {
  const int tid = threadIdx.x;
  red[tid] = v;
  __syncthreads();
  for (int s = kRedThreads / 2; s > 0; s >>= 1) {
    if (tid < s) red[tid] += red[tid + s];
    __syncthreads();
  }
  const float r = red[0];
  __syncthreads();
  return r;
}

// loss = mean((rgb - px)^2) over B*3 (train.py:89); d_rgb = 2*(rgb-px)/(3B)
__global__ __launch_bounds__(kRedThreads) void mse_grad_kernel(const float* __restrict__ rgb,
                                                               const float* __restrict__ px, int64_t B,
                                                               float* __restrict__ d_rgb,
                                                               float* __restrict__ sse_out) {
  __shared__ float red[kRedThreads];
  const int64_t n = B * 3;
  const float scale = 2.f / (float)n;
  float s = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += kRedThreads) {
    const float diff = rgb[i] - px[i];
    d_rgb[i] = diff * scale;
    s += diff * diff;
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) sse_out[0] = s;
}

int launch_mse_grad(const float* rgb, const float* pixels, int64_t B, float* d_rgb, float* sse_out, hipStream_t s) {
  hipLaunchKernelGGL(mse_grad_kernel, dim3(1), dim3(kRedThreads), 0, s, rgb, pixels, B, d_rgb, sse_out);
  return check_launch("mse_grad");
}

// loss_sp = weight * (1 - mean(exp(-length * relu(sigma))))  (train.py:81-83)
__global__ __launch_bounds__(kRedThreads) void sparsity_grad_kernel(const float* __restrict__ raw_sigma, int64_t n,
                                                                    float weight, float length,
                                                                    float* __restrict__ d_raw_sigma,
                                                                    float* __restrict__ sum_exp_out) {
  __shared__ float red[kRedThreads];
  float s = 0.f;
  const float gscale = weight * length / (float)n;
  for (int64_t i = threadIdx.x; i < n; i += kRedThreads) {
    const float raw = raw_sigma[i];
    const float e = expf(-length * fmaxf(raw, 0.f));
    s += e;
    d_raw_sigma[i] = raw > 0.f ? gscale * e : 0.f;
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) sum_exp_out[0] = s;
}

int launch_sparsity_grad(const float* raw_sigma, int64_t n, float weight, float length, float* d_raw_sigma,
                         float* sum_exp_out, hipStream_t s) {
  hipLaunchKernelGGL(sparsity_grad_kernel, dim3(1), dim3(kRedThreads), 0, s, raw_sigma, n, weight, length,
                     d_raw_sigma, sum_exp_out);
  return check_launch("sparsity_grad");
}

// sum of squares in two fixed-order stages (weight_l2, train.py:101-108)
constexpr int kSumsqBlocks = 64;
__global__ __launch_bounds__(kRedThreads) void sumsq_stage1(const float* __restrict__ x, int64_t n,
                                                            float* __restrict__ partial) {
  __shared__ float red[kRedThreads];
  float s = 0.f;
  for (int64_t i = blockIdx.x * (int64_t)kRedThreads + threadIdx.x; i < n; i += (int64_t)kSumsqBlocks * kRedThreads)
    s += x[i] * x[i];
  s = block_sum(s, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
__global__ void sumsq_stage2(const float* __restrict__ partial, float* __restrict__ out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float s = 0.f;
    for (int i = 0; i < kSumsqBlocks; ++i) s += partial[i];
    out[0] = s;
  }
}
// out must hold 1 + kSumsqBlocks floats: out[0] = result, out[1..] scratch
int launch_sumsq(const float* x, int64_t n, float* out, hipStream_t s) {
  hipLaunchKernelGGL(sumsq_stage1, dim3(kSumsqBlocks), dim3(kRedThreads), 0, s, x, n, out + 1);
  hipLaunchKernelGGL(sumsq_stage2, dim3(1), dim3(64), 0, s, out + 1, out);
  return check_launch("sumsq");
}

// Stats (nerf_sh/nerf/utils.py:43-50): loss, psnr, loss_c, loss_sp, psnr_c, weight_l2
__global__ void finalize_stats_kernel(const float* sse_f, const float* sse_c, const float* sum_exp,
                                      const float* sumsq, int64_t B, int has_fine, int64_t n_sp,
                                      float sp_weight, int64_t n_params, float* stats) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const float inv = 1.f / (float)(B * 3);
  const float ln10 = 2.302585092994046f;
  const float loss_last = (has_fine ? sse_f[0] : sse_c[0]) * inv;
  stats[0] = loss_last;
  stats[1] = -10.f * logf(loss_last) / ln10;
  if (has_fine) {
    const float lc = sse_c[0] * inv;
    stats[2] = lc;
    stats[4] = -10.f * logf(lc) / ln10;
  } else {
    stats[2] = 0.f;
    stats[4] = 0.f;
  }
  stats[3] = (sp_weight > 0.f && n_sp > 0) ? sp_weight * (1.f - sum_exp[0] / (float)n_sp) : 0.f;
  stats[5] = sumsq[0] / (float)n_params;
}

int launch_finalize_stats(const float* sse_f, const float* sse_c, const float* sum_exp, const float* sumsq,
                          int64_t B, int has_fine, int64_t n_sp, float sp_weight, int64_t n_params, float* stats,
                          hipStream_t s) {
  hipLaunchKernelGGL(finalize_stats_kernel, dim3(1), dim3(64), 0, s, sse_f, sse_c, sum_exp, sumsq, B, has_fine,
                     n_sp, sp_weight, n_params, stats);
  return check_launch("finalize_stats");
}

__global__ void fill_kernel(float* __restrict__ p, int64_t n, float v) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}
int launch_fill(float* p, int64_t n, float v, hipStream_t s) {
  if (n == 0) return PXO_OK;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(fill_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p, n, v);
  return check_launch("fill");
}

// y += a * x  (weight-decay term of loss_fn, train.py:101-114: d/dp [wd * sum(p^2)/n] = 2 wd p / n)
__global__ void axpy_kernel(float* __restrict__ y, const float* __restrict__ x, int64_t n, float a) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    y[i] = fmaf(a, x[i], y[i]);
}
int launch_axpy(float* y, const float* x, int64_t n, float a, hipStream_t s) {
  if (n == 0) return PXO_OK;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(axpy_kernel, dim3((unsigned)blocks), dim3(256), 0, s, y, x, n, a);
  return check_launch("axpy");
}

// flax.optim.Adam.apply_param_gradient with weight_decay = 0:
//   m = b1*m + (1-b1)*g ; v = b2*v + (1-b2)*g^2 ; t = step+1
//   p -= lr * (m/(1-b1^t)) / (sqrt(v/(1-b2^t)) + eps)
__global__ void adam_kernel(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                            const float* __restrict__ g, int64_t n, float lr, float bc1, float bc2,
                            float grad_scale) {
  // (1. - beta) is a python double in flax, rounded to f32 when it meets the f32 gradient
  const float b1 = 0.9f, b2 = 0.999f, eps = 1e-8f, omb1 = (float)(1.0 - 0.9), omb2 = (float)(1.0 - 0.999);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float gi = g[i] * grad_scale;
    const float mi = b1 * m[i] + omb1 * gi;
    const float vi = b2 * v[i] + omb2 * (gi * gi);
    m[i] = mi;
    v[i] = vi;
    const float mhat = mi / bc1, vhat = vi / bc2;
    p[i] = p[i] - lr * mhat / (sqrtf(vhat) + eps);
  }
}

int launch_adam(float* p, float* m, float* v, const float* g, int64_t n, float lr, int64_t step, float grad_scale,
                hipStream_t s) {
  if (n == 0) return PXO_OK;
  const double t = (double)(step + 1);
  const float bc1 = (float)(1.0 - pow(0.9, t)), bc2 = (float)(1.0 - pow(0.999, t));
  int64_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p, m, v, g, n, lr, bc1, bc2, grad_scale);
  return check_launch("adam");
}

}  // namespace pxo
