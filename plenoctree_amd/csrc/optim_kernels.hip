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

// Stats (nerf_sh/nerf/utils.py:43-50): loss, psnr, loss_c, loss_sp, psnr_c, weight_l2 from the per-ray squared
// errors of the two passes (shade_composite_train_kernel), the per-point exp(-len relu(sigma)) of the sparsity rows
// and the partial sums of squares of the parameters -- every sum in a fixed order (deterministic).
__global__ __launch_bounds__(kRedThreads) void finalize_stats_kernel(const float* __restrict__ sse_f,
                                                                     const float* __restrict__ sse_c,
                                                                     const float* __restrict__ sp_exp,
                                                                     const float* __restrict__ sumsq_partial, int64_t B,
                                                                     int64_t n_sp, float sp_weight, int64_t n_params,
                                                                     float* __restrict__ stats) {
  __shared__ float red[kRedThreads];
  auto total = [&](const float* x, int64_t n) {
    float s = 0.f;
    if (x) for (int64_t i = threadIdx.x; i < n; i += kRedThreads) s += x[i];
    return block_sum(s, red);
  };
  const float tf = total(sse_f, B), tc = total(sse_c, B), te = total(sp_exp, n_sp);
  const float tq = total(sumsq_partial, kSumsqBlocks);
  if (threadIdx.x != 0) return;
  const float inv = 1.f / (float)(B * 3);
  const float ln10 = 2.302585092994046f;
  const bool has_fine = sse_f != nullptr;
  const float loss_last = (has_fine ? tf : tc) * inv;
  stats[0] = loss_last;
  stats[1] = -10.f * logf(loss_last) / ln10;
  if (has_fine) {
    const float lc = tc * inv;
    stats[2] = lc;
    stats[4] = -10.f * logf(lc) / ln10;
  } else {
    stats[2] = 0.f;
    stats[4] = 0.f;
  }
  stats[3] = (sp_weight > 0.f && n_sp > 0) ? sp_weight * (1.f - te / (float)n_sp) : 0.f;
  stats[5] = tq / (float)n_params;
}

int launch_finalize_stats(const float* sse_f, const float* sse_c, const float* sp_exp, const float* sumsq_partial,
                          int64_t B, int64_t n_sp, float sp_weight, int64_t n_params, float* stats, hipStream_t s) {
  hipLaunchKernelGGL(finalize_stats_kernel, dim3(1), dim3(kRedThreads), 0, s, sse_f, sse_c, sp_exp, sumsq_partial, B,
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
// One rounding per operation, in this order (no fused multiply-add: the expression means the same in every kernel that
// inlines it, so pxo_adam_step and pxo_adam_pack_step agree bit for bit, and with an element-wise float32 evaluation).
__device__ __forceinline__ float adam_update(float p, float& m, float& v, float g, float lr, float bc1, float bc2) {
#pragma clang fp contract(off)    // (HIP's __fmul_rn / __fadd_rn are plain operators: they do not stop the contraction)
  // (1. - beta) is a python double in flax, rounded to f32 when it meets the f32 gradient
  const float b1 = 0.9f, b2 = 0.999f, eps = 1e-8f, omb1 = (float)(1.0 - 0.9), omb2 = (float)(1.0 - 0.999);
  const float mi = b1 * m + omb1 * g;
  const float vi = b2 * v + omb2 * (g * g);
  m = mi;
  v = vi;
  const float mhat = mi / bc1, vhat = vi / bc2;
  return p - lr * mhat / (sqrtf(vhat) + eps);
}

__global__ void adam_kernel(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                            const float* __restrict__ g, int64_t n, float lr, float bc1, float bc2,
                            float grad_scale) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float mi = m[i], vi = v[i];
    p[i] = adam_update(p[i], mi, vi, g[i] * grad_scale, lr, bc1, bc2);
    m[i] = mi;
    v[i] = vi;
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

// Adam and the refresh of the four MFMA-fragment-ordered weight images in ONE launch: the thread that updates
// parameter i also stores it at its place in the forward image and (for the entries the backward(data) GEMMs use: the
// first 256 input rows of Dense_1..7 and the heads) in the transposed image.  Zero padding of the images is written once
// by pxo_pack_weights and never touched here.  5 launches per step -> 1 (the step at 512 rays per GPU is launch-bound
// around its small kernels).
struct LeafOffsets { int64_t kernel[11]; };     // kernel[l] = leaf_kernel_off(l), kernel[10] = floats per MLP

__global__ void adam_pack_kernel(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                                 const float* __restrict__ g, int64_t n_mlp, int deg, LeafOffsets lo, float lr, float bc1,
                                 float bc2, float grad_scale, float* __restrict__ fwd0, float* __restrict__ bwd0,
                                 float* __restrict__ fwd1, float* __restrict__ bwd1) {
  const int C = rgb_channels(deg), nhb = head_blocks(deg);
  const int64_t bias0 = fwd_bias_off(deg), head0 = fwd_layer_off(8);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < 2 * n_mlp; i += (int64_t)gridDim.x * blockDim.x) {
    float mi = m[i], vi = v[i];
    const float pn = adam_update(p[i], mi, vi, g[i] * grad_scale, lr, bc1, bc2);
    m[i] = mi;
    v[i] = vi;
    p[i] = pn;
    const bool second = i >= n_mlp;
    const int64_t j = second ? i - n_mlp : i;
    float* __restrict__ fwd = second ? fwd1 : fwd0;
    float* __restrict__ bwd = second ? bwd1 : bwd0;
    int l = 0;
#pragma unroll
    for (int q = 1; q < 10; ++q) l += (j >= lo.kernel[q]) ? 1 : 0;
    const int nout = layer_out(l, deg);
    const int64_t rel = j - lo.kernel[l];
    const int64_t nk = (int64_t)layer_in(l) * nout;
    if (rel >= nk) {                                   // bias
      const int n = (int)(rel - nk);
      fwd[bias0 + (l < 8 ? l * kW + n : 8 * kW + (l == 9 ? n : C))] = pn;
      continue;
    }
    const int k = (int)(rel / nout), n = (int)(rel - (int64_t)k * nout);
    if (l < 8) {
      fwd[fwd_layer_off(l) + packed_index(k, n, 8)] = pn;
      if (bwd && l >= 1 && k < kW) bwd[bwd_layer_off(l, deg) + packed_index(n, k, 8)] = pn;
    } else {
      const int col = l == 9 ? n : C;                  // fused head: [Dense_9 | Dense_8]
      fwd[head0 + packed_index(k, col, nhb)] = pn;
      if (bwd) bwd[packed_index(col, k, 8)] = pn;
    }
  }
}

int launch_adam_pack(const PxoCfg* cfg, float* p, float* m, float* v, const float* g, float lr, int64_t step,
                     float grad_scale, float* fwd0, float* bwd0, float* fwd1, float* bwd1, hipStream_t s) {
  const int deg = cfg->sh_deg;
  const int64_t n_mlp = mlp_param_count(deg);
  LeafOffsets lo;
  for (int l = 0; l <= 10; ++l) lo.kernel[l] = leaf_kernel_off(l, deg);
  const double t = (double)(step + 1);
  const float bc1 = (float)(1.0 - pow(0.9, t)), bc2 = (float)(1.0 - pow(0.999, t));
  int64_t blocks = (2 * n_mlp + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(adam_pack_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p, m, v, g, n_mlp, deg, lo, lr, bc1, bc2,
                     grad_scale, fwd0, bwd0, fwd1, bwd1);
  return check_launch("adam_pack");
}

}  // namespace pxo
