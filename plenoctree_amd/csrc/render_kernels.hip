// Wavefront-parallel ray kernels for gfx950: stratified sampling, SH shading + alpha compositing
// (forward and reverse), hierarchical PDF resampling + sort, and the counter-based uniform
// generator.  One 64-lane wave per ray; compositing uses a wave product-scan (DPP shuffles).
//
// Reference semantics: nerf_sh/nerf/model_utils.py:97-314, nerf_sh/nerf/sh.py:54-109,
// nerf_sh/nerf/models.py:269-307.
#include "pxo_common.h"
#include "pxo_sh.h"

namespace pxo {

constexpr int kRayThreads = 256;             // 4 rays per workgroup
constexpr int kRaysPerBlock = kRayThreads / 64;

// torch/jnp.linspace(start, end, n)[i] in float32 (symmetric evaluation like ATen's kernel)
__device__ __forceinline__ float linspace_at(float start, float end, int n, int i) {
  if (n <= 1) return start;
  const float step = (end - start) / (float)(n - 1);
  return i < n / 2 ? start + step * (float)i : end - step * (float)(n - 1 - i);
}

// ------------------------------------------------------------------------------------------
// sample_along_rays + cast_rays (model_utils.py:104-142, :97-101)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float coarse_z(int s, int S, float near_, float far_, int lindisp) {
  const float t = linspace_at(0.f, 1.f, S, s);
  if (lindisp) return 1.f / (1.f / near_ * (1.f - t) + 1.f / far_ * t);
  return near_ * (1.f - t) + far_ * t;
}

__global__ void sample_along_rays_kernel(const float* __restrict__ o, const float* __restrict__ d,
                                         int64_t B, int S, float near_, float far_, int lindisp,
                                         const float* __restrict__ t_rand, float* __restrict__ z_out,
                                         float* __restrict__ pts) {
  const int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (idx >= B * S) return;
  const int64_t b = idx / S;
  const int s = (int)(idx - b * S);
  float z = coarse_z(s, S, near_, far_, lindisp);
  if (t_rand) {
    const float lower = s == 0 ? z : 0.5f * (z + coarse_z(s - 1, S, near_, far_, lindisp));
    const float upper = s == S - 1 ? z : 0.5f * (coarse_z(s + 1, S, near_, far_, lindisp) + z);
    z = lower + (upper - lower) * t_rand[idx];
  }
  z_out[idx] = z;
  pts[idx * 3 + 0] = o[b * 3 + 0] + z * d[b * 3 + 0];
  pts[idx * 3 + 1] = o[b * 3 + 1] + z * d[b * 3 + 1];
  pts[idx * 3 + 2] = o[b * 3 + 2] + z * d[b * 3 + 2];
}

int launch_sample_along_rays(const float* o, const float* d, int64_t B, int S, float near_, float far_,
                             int lindisp, const float* t_rand, float* z, float* pts, hipStream_t s) {
  if (B == 0) return PXO_OK;
  const int64_t n = B * S;
  hipLaunchKernelGGL(sample_along_rays_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, o, d, B, S,
                     near_, far_, lindisp, t_rand, z, pts);
  return check_launch("sample_along_rays");
}

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  return v;
}
// inclusive product scan across the 64 lanes
__device__ __forceinline__ float wave_scan_mul(float v, int lane) {
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const float t = __shfl_up(v, off);
    if (lane >= off) v *= t;
  }
  return v;
}
// inclusive sum scan towards lane 0 (suffix sums)
__device__ __forceinline__ float wave_rscan_add(float v, int lane) {
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const float t = __shfl_down(v, off);
    if (lane + off < 64) v += t;
  }
  return v;
}

constexpr int kMaxChunks = 4;  // up to 256 samples per ray

// per-lane state of one 64-sample chunk of a ray
struct SampleState {
  float rgb[3];
  float e;      // exp(-sigma*dist)
  float T;      // transmittance before this sample
  float z;
  float dist;
  float raw_sigma;
};

// loads the chunk's raw SH coefficients through LDS (coalesced), evaluates sigmoid(eval_sh),
// relu(sigma), alpha and the transmittance scan; `carry` is the product over previous chunks.
template <int DEG>
__device__ __forceinline__ void shade_chunk(const float* __restrict__ raw_rgb, const float* __restrict__ raw_sigma,
                                            const float* __restrict__ z_vals, float* __restrict__ wl,
                                            const float (&Y)[(DEG + 1) * (DEG + 1)], int64_t ray, int S,
                                            int ch, int lane, float norm_d, float& carry, SampleState& st) {
  constexpr int K = (DEG + 1) * (DEG + 1), C = 3 * K, CS = C | 1;
  const int s0 = ch * 64;
  const int nvalid = S - s0 < 64 ? S - s0 : 64;
  const int64_t base = ray * S + s0;
  __syncthreads();
  for (int idx = lane; idx < nvalid * C; idx += 64) {
    const int s = idx / C, j = idx - s * C;
    wl[s * CS + j] = raw_rgb[base * C + idx];
  }
  __syncthreads();
  const bool valid = lane < nvalid;
  float f = 1.f;
  st.e = 1.f; st.z = 0.f; st.dist = 0.f; st.raw_sigma = 0.f;
  st.rgb[0] = st.rgb[1] = st.rgb[2] = 0.f;
  if (valid) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float pre = 0.f;
#pragma unroll
      for (int k = 0; k < K; ++k) pre += Y[k] * wl[lane * CS + c * K + k];
      st.rgb[c] = 1.f / (1.f + expf(-pre));        // sigmoid, models.py:280
    }
    st.raw_sigma = raw_sigma[base + lane];
    const float sigma = fmaxf(st.raw_sigma, 0.f);  // relu, models.py:281
    st.z = z_vals[base + lane];
    const bool last = (s0 + lane == S - 1);
    st.dist = (last ? 1e10f : z_vals[base + lane + 1] - st.z) * norm_d;
    st.e = expf(-sigma * st.dist);
    f = (1.f - (1.f - st.e)) + 1e-10f;             // 1 - alpha + eps, model_utils.py:202
  }
  const float incl = wave_scan_mul(f, lane);
  float excl = __shfl_up(incl, 1);
  if (lane == 0) excl = 1.f;
  st.T = carry * excl;
  carry = carry * __shfl(incl, 63);
}

template <int DEG>
__global__ __launch_bounds__(kRayThreads) void shade_composite_fwd_kernel(
    const float* __restrict__ raw_rgb, const float* __restrict__ raw_sigma, const float* __restrict__ z_vals,
    const float* __restrict__ dirs, const float* __restrict__ viewdirs, int64_t B, int S, int white,
    float* __restrict__ comp_rgb, float* __restrict__ disp, float* __restrict__ acc_out,
    float* __restrict__ weights) {
  constexpr int K = (DEG + 1) * (DEG + 1), C = 3 * K, CS = C | 1;
  __shared__ float lds[kRaysPerBlock][64 * CS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int64_t ray = blockIdx.x * (int64_t)kRaysPerBlock + wave;
  const bool ray_ok = ray < B;
  if (!ray_ok) ray = B - 1;
  float Y[K];
  sh_basis<DEG>(viewdirs[ray * 3], viewdirs[ray * 3 + 1], viewdirs[ray * 3 + 2], Y);
  const float dx = dirs[ray * 3], dy = dirs[ray * 3 + 1], dz = dirs[ray * 3 + 2];
  const float norm_d = sqrtf(dx * dx + dy * dy + dz * dz);
  float carry = 1.f, s_r = 0.f, s_g = 0.f, s_b = 0.f, s_depth = 0.f, s_acc = 0.f;
  const int nch = (S + 63) / 64;
  for (int ch = 0; ch < nch; ++ch) {
    SampleState st;
    shade_chunk<DEG>(raw_rgb, raw_sigma, z_vals, lds[wave], Y, ray, S, ch, lane, norm_d, carry, st);
    const float w = (1.f - st.e) * st.T;
    s_r += w * st.rgb[0]; s_g += w * st.rgb[1]; s_b += w * st.rgb[2];
    s_depth += w * st.z; s_acc += w;
    if (ray_ok && ch * 64 + lane < S) weights[ray * S + ch * 64 + lane] = w;
  }
  s_r = wave_sum(s_r); s_g = wave_sum(s_g); s_b = wave_sum(s_b);
  s_depth = wave_sum(s_depth); s_acc = wave_sum(s_acc);
  if (ray_ok && lane == 0) {
    const float inv_eps = 1e10f;
    float dsp = s_acc / s_depth;
    dsp = (dsp > 0.f && dsp < inv_eps && s_acc > 1e-10f) ? dsp : inv_eps;  // model_utils.py:217-219
    const float bg = white ? 1.f - s_acc : 0.f;
    comp_rgb[ray * 3 + 0] = s_r + bg;
    comp_rgb[ray * 3 + 1] = s_g + bg;
    comp_rgb[ray * 3 + 2] = s_b + bg;
    disp[ray] = dsp;
    acc_out[ray] = s_acc;
  }
}

// reverse of the above for a loss on comp_rgb only; no gradient flows to z (stop_gradient,
// model_utils.py:286, and the stratified z are parameter-free).
template <int DEG>
__global__ __launch_bounds__(kRayThreads) void shade_composite_bwd_kernel(
    const float* __restrict__ raw_rgb, const float* __restrict__ raw_sigma, const float* __restrict__ z_vals,
    const float* __restrict__ dirs, const float* __restrict__ viewdirs, const float* __restrict__ d_comp,
    int64_t B, int S, int white, float* __restrict__ d_raw_rgb, float* __restrict__ d_raw_sigma) {
  constexpr int K = (DEG + 1) * (DEG + 1), C = 3 * K, CS = C | 1;
  __shared__ float lds[kRaysPerBlock][64 * CS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int64_t ray = blockIdx.x * (int64_t)kRaysPerBlock + wave;
  const bool ray_ok = ray < B;
  if (!ray_ok) ray = B - 1;
  float Y[K];
  sh_basis<DEG>(viewdirs[ray * 3], viewdirs[ray * 3 + 1], viewdirs[ray * 3 + 2], Y);
  const float dx = dirs[ray * 3], dy = dirs[ray * 3 + 1], dz = dirs[ray * 3 + 2];
  const float norm_d = sqrtf(dx * dx + dy * dy + dz * dz);
  const float g0 = d_comp[ray * 3], g1 = d_comp[ray * 3 + 1], g2 = d_comp[ray * 3 + 2];
  const float bgc = white ? 1.f : 0.f;
  const int nch = (S + 63) / 64;
  SampleState st[kMaxChunks];
  float carry = 1.f;
#pragma unroll
  for (int ch = 0; ch < kMaxChunks; ++ch)
    if (ch < nch) shade_chunk<DEG>(raw_rgb, raw_sigma, z_vals, lds[wave], Y, ray, S, ch, lane, norm_d, carry, st[ch]);
  float suffix = 0.f;  // sum over later samples of dL/dw_j * w_j
#pragma unroll
  for (int ch = kMaxChunks - 1; ch >= 0; --ch) {
    if (ch >= nch) continue;
    const SampleState& q = st[ch];
    const bool valid = ch * 64 + lane < S;
    const float alpha = 1.f - q.e;
    const float w = alpha * q.T;
    // comp = sum_s w_s c_s + bg*(1 - sum_s w_s)
    const float dw = g0 * (q.rgb[0] - bgc) + g1 * (q.rgb[1] - bgc) + g2 * (q.rgb[2] - bgc);
    const float G = valid ? dw * w : 0.f;
    const float incl = wave_rscan_add(G, lane);
    float excl = __shfl_down(incl, 1);
    if (lane == 63) excl = 0.f;
    const float R = suffix + excl;
    suffix += __shfl(incl, 0);
    const float fct = (1.f - alpha) + 1e-10f;
    const float dalpha = dw * q.T - R / fct;
    const float dsigma = dalpha * q.dist * q.e;     // d(1-exp(-s*dist))/ds
    const int64_t base = ray * S + ch * 64;
    if (ray_ok && valid) d_raw_sigma[base + lane] = q.raw_sigma > 0.f ? dsigma : 0.f;
    __syncthreads();
    if (valid) {
      const float dp[3] = {g0 * w * q.rgb[0] * (1.f - q.rgb[0]), g1 * w * q.rgb[1] * (1.f - q.rgb[1]),
                           g2 * w * q.rgb[2] * (1.f - q.rgb[2])};
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int k = 0; k < K; ++k) lds[wave][lane * CS + c * K + k] = dp[c] * Y[k];
    }
    __syncthreads();
    const int nvalid = S - ch * 64 < 64 ? S - ch * 64 : 64;
    if (ray_ok)
      for (int idx = lane; idx < nvalid * C; idx += 64) {
        const int s = idx / C, j = idx - s * C;
        d_raw_rgb[base * C + idx] = lds[wave][s * CS + j];
      }
  }
}

// Training form: forward compositing, the pixel loss of nerf_sh/train.py:89-98 and the reverse pass in ONE launch (the
// backward kernel above already re-derives the whole forward state in registers; the loss gradient of a ray depends on
// that ray's colour only).  Per ray: comp_rgb (+weights for sample_pdf), sse[ray] = sum_c (comp_c - px_c)^2 (summed
// in a fixed order by finalize_stats), d_comp = 2 (comp - px) / (3 B), then d_raw_rgb / d_raw_sigma as above.
// Blocks past the rays serve the sparsity rows appended to the pass (train.py:77-85): e = exp(-len relu(s)),
// d_raw_sigma = w len e / n for s > 0, and a zero gradient on their raw_rgb.
template <int DEG>
__global__ __launch_bounds__(kRayThreads) void shade_composite_train_kernel(
    const float* __restrict__ raw_rgb, const float* __restrict__ raw_sigma, const float* __restrict__ z_vals,
    const float* __restrict__ dirs, const float* __restrict__ viewdirs, const float* __restrict__ pixels, int64_t B,
    int S, int white, float* __restrict__ comp_rgb, float* __restrict__ weights, float* __restrict__ ray_sse,
    float* __restrict__ d_raw_rgb, float* __restrict__ d_raw_sigma, int64_t n_sp, float sp_weight, float sp_length,
    float* __restrict__ sp_exp) {
  constexpr int K = (DEG + 1) * (DEG + 1), C = 3 * K, CS = C | 1;
  __shared__ float lds[kRaysPerBlock][64 * CS];
  const int64_t ray_blocks = (B + kRaysPerBlock - 1) / kRaysPerBlock;
  if ((int64_t)blockIdx.x >= ray_blocks) {       // sparsity rows [B*S, B*S + n_sp)
    const int64_t r0 = ((int64_t)blockIdx.x - ray_blocks) * kRayThreads;
    const int64_t row = r0 + threadIdx.x;
    const int64_t base = B * S;
    if (row < n_sp) {
      const float raw = raw_sigma[base + row];
      const float e = expf(-sp_length * fmaxf(raw, 0.f));
      sp_exp[row] = e;
      d_raw_sigma[base + row] = raw > 0.f ? (sp_weight * sp_length / (float)n_sp) * e : 0.f;
    }
    const int64_t nrow = n_sp - r0 < kRayThreads ? n_sp - r0 : kRayThreads;
    float* __restrict__ z0 = d_raw_rgb + (base + r0) * C;
    for (int64_t i = threadIdx.x; i < nrow * C; i += kRayThreads) z0[i] = 0.f;
    return;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int64_t ray = blockIdx.x * (int64_t)kRaysPerBlock + wave;
  const bool ray_ok = ray < B;
  if (!ray_ok) ray = B - 1;
  float Y[K];
  sh_basis<DEG>(viewdirs[ray * 3], viewdirs[ray * 3 + 1], viewdirs[ray * 3 + 2], Y);
  const float dx = dirs[ray * 3], dy = dirs[ray * 3 + 1], dz = dirs[ray * 3 + 2];
  const float norm_d = sqrtf(dx * dx + dy * dy + dz * dz);
  const float bgc = white ? 1.f : 0.f;
  const int nch = (S + 63) / 64;
  SampleState st[kMaxChunks];
  float carry = 1.f, s_r = 0.f, s_g = 0.f, s_b = 0.f, s_acc = 0.f;
#pragma unroll
  for (int ch = 0; ch < kMaxChunks; ++ch)
    if (ch < nch) {
      shade_chunk<DEG>(raw_rgb, raw_sigma, z_vals, lds[wave], Y, ray, S, ch, lane, norm_d, carry, st[ch]);
      const float w = (1.f - st[ch].e) * st[ch].T;
      s_r += w * st[ch].rgb[0]; s_g += w * st[ch].rgb[1]; s_b += w * st[ch].rgb[2]; s_acc += w;
      if (weights && ray_ok && ch * 64 + lane < S) weights[ray * S + ch * 64 + lane] = w;
    }
  s_r = wave_sum(s_r); s_g = wave_sum(s_g); s_b = wave_sum(s_b); s_acc = wave_sum(s_acc);
  const float bg = white ? 1.f - s_acc : 0.f;
  const float c0 = s_r + bg, c1 = s_g + bg, c2 = s_b + bg;
  const float e0 = c0 - pixels[ray * 3], e1 = c1 - pixels[ray * 3 + 1], e2 = c2 - pixels[ray * 3 + 2];
  const float scale = 2.f / (float)(B * 3);
  const float g0 = e0 * scale, g1 = e1 * scale, g2 = e2 * scale;
  if (ray_ok && lane == 0) {
    ray_sse[ray] = (e0 * e0 + e1 * e1) + e2 * e2;
    if (comp_rgb) { comp_rgb[ray * 3] = c0; comp_rgb[ray * 3 + 1] = c1; comp_rgb[ray * 3 + 2] = c2; }
  }
  float suffix = 0.f;  // sum over later samples of dL/dw_j * w_j
#pragma unroll
  for (int ch = kMaxChunks - 1; ch >= 0; --ch) {
    if (ch >= nch) continue;
    const SampleState& q = st[ch];
    const bool valid = ch * 64 + lane < S;
    const float alpha = 1.f - q.e;
    const float w = alpha * q.T;
    const float dw = g0 * (q.rgb[0] - bgc) + g1 * (q.rgb[1] - bgc) + g2 * (q.rgb[2] - bgc);
    const float G = valid ? dw * w : 0.f;
    const float incl = wave_rscan_add(G, lane);
    float excl = __shfl_down(incl, 1);
    if (lane == 63) excl = 0.f;
    const float R = suffix + excl;
    suffix += __shfl(incl, 0);
    const float fct = (1.f - alpha) + 1e-10f;
    const float dalpha = dw * q.T - R / fct;
    const float dsigma = dalpha * q.dist * q.e;
    const int64_t base = ray * S + ch * 64;
    if (ray_ok && valid) d_raw_sigma[base + lane] = q.raw_sigma > 0.f ? dsigma : 0.f;
    __syncthreads();
    if (valid) {
      const float dp[3] = {g0 * w * q.rgb[0] * (1.f - q.rgb[0]), g1 * w * q.rgb[1] * (1.f - q.rgb[1]),
                           g2 * w * q.rgb[2] * (1.f - q.rgb[2])};
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int k = 0; k < K; ++k) lds[wave][lane * CS + c * K + k] = dp[c] * Y[k];
    }
    __syncthreads();
    const int nvalid = S - ch * 64 < 64 ? S - ch * 64 : 64;
    if (ray_ok)
      for (int idx = lane; idx < nvalid * C; idx += 64) {
        const int s = idx / C, j = idx - s * C;
        d_raw_rgb[base * C + idx] = lds[wave][s * CS + j];
      }
  }
}

#define PXO_DEG_SWITCH(deg, CALL) \
  switch (deg) {                  \
    case 0: CALL(0); break;       \
    case 1: CALL(1); break;       \
    case 2: CALL(2); break;       \
    case 3: CALL(3); break;       \
    default: CALL(4); break;      \
  }

int launch_shade_composite_fwd(const PxoCfg* cfg, const float* raw_rgb, const float* raw_sigma, const float* z,
                               const float* dirs, const float* viewdirs, int64_t B, int S, float* comp_rgb,
                               float* disp, float* acc, float* weights, hipStream_t s) {
  if (B == 0) return PXO_OK;
  if (S > 64 * kMaxChunks || S < 1) { set_error("samples per ray %d not in [1,%d]", S, 64 * kMaxChunks); return PXO_ERR_ARG; }
  dim3 grid((unsigned)((B + kRaysPerBlock - 1) / kRaysPerBlock)), block(kRayThreads);
#define CALL(D) hipLaunchKernelGGL((shade_composite_fwd_kernel<D>), grid, block, 0, s, raw_rgb, raw_sigma, z, dirs, \
                                   viewdirs, B, S, cfg->white_bkgd, comp_rgb, disp, acc, weights)
  PXO_DEG_SWITCH(cfg->sh_deg, CALL)
#undef CALL
  return check_launch("shade_composite_fwd");
}

int launch_shade_composite_bwd(const PxoCfg* cfg, const float* raw_rgb, const float* raw_sigma, const float* z,
                               const float* dirs, const float* viewdirs, const float* d_comp_rgb, int64_t B, int S,
                               float* d_raw_rgb, float* d_raw_sigma, hipStream_t s) {
  if (B == 0) return PXO_OK;
  if (S > 64 * kMaxChunks || S < 1) { set_error("samples per ray %d not in [1,%d]", S, 64 * kMaxChunks); return PXO_ERR_ARG; }
  dim3 grid((unsigned)((B + kRaysPerBlock - 1) / kRaysPerBlock)), block(kRayThreads);
#define CALL(D) hipLaunchKernelGGL((shade_composite_bwd_kernel<D>), grid, block, 0, s, raw_rgb, raw_sigma, z, dirs, \
                                   viewdirs, d_comp_rgb, B, S, cfg->white_bkgd, d_raw_rgb, d_raw_sigma)
  PXO_DEG_SWITCH(cfg->sh_deg, CALL)
#undef CALL
  return check_launch("shade_composite_bwd");
}

int launch_shade_composite_train(const PxoCfg* cfg, const float* raw_rgb, const float* raw_sigma, const float* z,
                                 const float* dirs, const float* viewdirs, const float* pixels, int64_t B, int S,
                                 float* comp_rgb, float* weights, float* ray_sse, float* d_raw_rgb, float* d_raw_sigma,
                                 int64_t n_sp, float* sp_exp, hipStream_t s) {
  if (B == 0) return PXO_OK;
  if (S > 64 * kMaxChunks || S < 1) { set_error("samples per ray %d not in [1,%d]", S, 64 * kMaxChunks); return PXO_ERR_ARG; }
  const int64_t blocks = (B + kRaysPerBlock - 1) / kRaysPerBlock + (n_sp + kRayThreads - 1) / kRayThreads;
  dim3 grid((unsigned)blocks), block(kRayThreads);
#define CALL(D) hipLaunchKernelGGL((shade_composite_train_kernel<D>), grid, block, 0, s, raw_rgb, raw_sigma, z, dirs, \
                                   viewdirs, pixels, B, S, cfg->white_bkgd, comp_rgb, weights, ray_sse, d_raw_rgb,       \
                                   d_raw_sigma, n_sp, cfg->sparsity_weight, cfg->sparsity_length, sp_exp)
  PXO_DEG_SWITCH(cfg->sh_deg, CALL)
#undef CALL
  return check_launch("shade_composite_train");
}

// ------------------------------------------------------------------------------------------
// sample_pdf: piecewise_constant_pdf + sort + cast_rays (model_utils.py:225-314), with
// bins = mid-points of z_coarse and weights = w_coarse[1:-1] (models.py:296-301)
// ------------------------------------------------------------------------------------------
constexpr int kMaxCoarse = 128, kMaxFine = 256;

__global__ __launch_bounds__(kRayThreads) void sample_pdf_kernel(
    const float* __restrict__ z_c, const float* __restrict__ w_c, const float* __restrict__ o,
    const float* __restrict__ d, int64_t B, int Nc, int Nf, const float* __restrict__ u_in,
    float* __restrict__ z_out, float* __restrict__ pts) {
  __shared__ float s_cdf[kRaysPerBlock][kMaxCoarse];
  __shared__ float s_bins[kRaysPerBlock][kMaxCoarse];
  __shared__ double s_pdf[kRaysPerBlock][kMaxCoarse];
  __shared__ float s_z[kRaysPerBlock][kMaxCoarse + kMaxFine];
  __shared__ float s_sorted[kRaysPerBlock][kMaxCoarse + kMaxFine];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int64_t ray = blockIdx.x * (int64_t)kRaysPerBlock + wave;
  const bool ray_ok = ray < B;
  if (!ray_ok) ray = B - 1;
  const int nb = Nc - 1;   // knots
  const int nw = Nc - 2;   // weights / bins between knots
  float* cdf = s_cdf[wave]; float* bins = s_bins[wave]; double* pdf = s_pdf[wave];
  float* zall = s_z[wave]; float* sorted = s_sorted[wave];

  // The weight sum, the pdf and the running cdf are accumulated in float64 and each knot is rounded to float32
  // once.  In float32 the cdf of a ray that ends on an opaque surface plateaus at 1 - 2^-23 or 1 - 2^-24 depending
  // on the last ulp of ~60 roundings, and 1 - 2^-23 is exactly the last deterministic sample u = 1 - eps
  // (model_utils.py:265): `u >= cdf` (:270) then moves that sample from the surface bin to the far end of the
  // plateau -- one bin width (6e-2) in z and up to 7e-3 in the pixel, for a handful of rays per 4096
  // (profiles/r03_render_outliers.md).  The exact sum is 1 - O(1e-16), so the float64 oracle never takes that
  // branch; accumulating in float64 removes the coin toss (and costs nothing: ~60 adds per ray).
  double wsum = 0.0;
  for (int i = lane; i < Nc; i += 64) {
    const float zi = z_c[ray * Nc + i];
    zall[i] = zi;
    if (i + 1 < Nc) bins[i] = 0.5f * (z_c[ray * Nc + i + 1] + zi);   // models.py:296
    if (i >= 1 && i <= nw) { const double wi = (double)w_c[ray * Nc + i]; pdf[i - 1] = wi; wsum += wi; }
  }
  wsum = wave_sum(wsum);
  // model_utils.py:240-244: pad so that the sum is at least eps
  const double padding = fmax(0.0, (double)1e-5f - wsum);
  const double wtot = wsum + padding;
  __syncthreads();
  for (int i = lane; i < nw; i += 64) pdf[i] = (pdf[i] + padding / (double)nw) / wtot;
  __syncthreads();
  if (lane == 0) {  // sequential cumsum keeps the cdf monotone (model_utils.py:248-257)
    cdf[0] = 0.f;
    double run = 0.0;
    for (int i = 0; i + 1 < nw; ++i) { run += pdf[i]; cdf[i + 1] = (float)fmin(1.0, run); }
    cdf[nw] = 1.f;
  }
  __syncthreads();
  const float u_hi = 1.f - 1.1920928955078125e-07f;   // 1 - finfo(float32).eps, model_utils.py:265
  for (int f = lane; f < Nf; f += 64) {
    const float u = u_in ? u_in[ray * Nf + f] : linspace_at(0.f, u_hi, Nf, f);
    int cnt = 0;
    for (int k = 0; k < nb; ++k) cnt += (u >= cdf[k]) ? 1 : 0;      // mask = u >= cdf (:270)
    int k0 = cnt - 1; if (k0 < 0) k0 = 0;
    int k1 = cnt < nb ? cnt : nb - 1;
    const float c0 = cdf[k0], c1 = cdf[k1], b0 = bins[k0], b1 = bins[k1];
    float t = (u - c0) / (c1 - c0);
    if (t != t) t = 0.f;                                            // nan_to_num (:282)
    t = fminf(fmaxf(t, 0.f), 1.f);
    zall[Nc + f] = b0 + t * (b1 - b0);
  }
  __syncthreads();
  const int n = Nc + Nf;
  for (int e = lane; e < n; e += 64) {   // rank sort: a sort is a permutation, values stay exact
    const float x = zall[e];
    int rank = 0;
    for (int j = 0; j < n; ++j) {
      const float y = zall[j];
      rank += (y < x || (y == x && j < e)) ? 1 : 0;
    }
    sorted[rank] = x;
  }
  __syncthreads();
  if (ray_ok) {
    const float ox = o[ray * 3], oy = o[ray * 3 + 1], oz = o[ray * 3 + 2];
    const float dx = d[ray * 3], dy = d[ray * 3 + 1], dz = d[ray * 3 + 2];
    for (int e = lane; e < n; e += 64) {
      const float z = sorted[e];
      const int64_t idx = ray * n + e;
      z_out[idx] = z;
      pts[idx * 3 + 0] = ox + z * dx;
      pts[idx * 3 + 1] = oy + z * dy;
      pts[idx * 3 + 2] = oz + z * dz;
    }
  }
}

int launch_sample_pdf(const float* z_c, const float* w_c, const float* o, const float* d, int64_t B, int Nc,
                      int Nf, const float* u, float* z_out, float* pts, hipStream_t s) {
  if (B == 0) return PXO_OK;
  if (Nc < 3 || Nc > kMaxCoarse || Nf < 1 || Nf > kMaxFine) {
    set_error("sample_pdf: Nc=%d (3..%d) Nf=%d (1..%d)", Nc, kMaxCoarse, Nf, kMaxFine);
    return PXO_ERR_ARG;
  }
  dim3 grid((unsigned)((B + kRaysPerBlock - 1) / kRaysPerBlock)), block(kRayThreads);
  hipLaunchKernelGGL(sample_pdf_kernel, grid, block, 0, s, z_c, w_c, o, d, B, Nc, Nf, u, z_out, pts);
  return check_launch("sample_pdf");
}

// ------------------------------------------------------------------------------------------
// generate_rays for a batch of pixels of one camera (nerf_sh/nerf/utils.py:545-589, pinhole):
// pixel id p -> x = p % W, y = p / W (integer pixel centres), dir_cam = [(x-W/2)/f, -(y-H/2)/f, -1],
// direction = R dir_cam (not normalised), origin = c2w[:3,3], viewdir = direction / |direction|
// ------------------------------------------------------------------------------------------
// With n_cams > 1 the ids index the flattened [n_cams, H*W] ray table of the image_batching sampler
// (nerf_sh/nerf/datasets.py:137-141,152-157) and c2w is [n_cams,3,4].
// ray of pixel p of the camera c2w, written to row i (generate_rays, nerf_sh/nerf/utils.py:545-589, pinhole branch)
__device__ __forceinline__ void pixel_ray(const float* __restrict__ c2w, int W, int H, float focal, int64_t p, int64_t i,
                                          float* __restrict__ o, float* __restrict__ d, float* __restrict__ v) {
  const float x = (float)(p % W), y = (float)(p / W);
  const float cx = (x - (float)W * 0.5f) / focal, cy = -(y - (float)H * 0.5f) / focal, cz = -1.f;
  float dir[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    // numpy matmul accumulates the 3 products in order
    dir[a] = c2w[a * 4 + 0] * cx + c2w[a * 4 + 1] * cy + c2w[a * 4 + 2] * cz;
    o[i * 3 + a] = c2w[a * 4 + 3];
    d[i * 3 + a] = dir[a];
  }
  const float n = sqrtf(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
#pragma unroll
  for (int a = 0; a < 3; ++a) v[i * 3 + a] = dir[a] / n;
}

__global__ void generate_rays_kernel(const float* __restrict__ c2w_all, int n_cams, int W, int H, float focal,
                                     const int64_t* __restrict__ pix, int64_t B, float* __restrict__ o,
                                     float* __restrict__ d, float* __restrict__ v) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= B) return;
  int64_t p = pix ? pix[i] : i;
  const float* __restrict__ c2w = c2w_all;
  if (n_cams > 1) {
    const int64_t hw = (int64_t)W * H;
    int64_t cam = p / hw;
    if (cam >= n_cams) cam = n_cams - 1;
    p -= cam * hw;
    c2w += cam * 12;
  }
  pixel_ray(c2w, W, H, focal, p, i, o, d, v);
}

int launch_generate_rays(const float* c2w, int n_cams, int W, int H, float focal, const int64_t* pix, int64_t B,
                         float* o, float* d, float* v, hipStream_t s) {
  if (B == 0) return PXO_OK;
  hipLaunchKernelGGL(generate_rays_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, s, c2w, n_cams, W, H, focal,
                     pix, B, o, d, v);
  return check_launch("generate_rays");
}

// uniform integers in [0, n) from the Philox stream (the role of np.random.randint in
// Dataset._next_train, nerf_sh/nerf/datasets.py:159-166)
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1);
__global__ void randint_kernel(uint64_t seed, uint64_t stream_id, int64_t count, int64_t n, int64_t* __restrict__ out);

// mean over the S samples of a leaf of cat([raw_rgb, raw_sigma]) (octree/extraction.py:391-393)
__global__ void mean_samples_kernel(const float* __restrict__ raw_rgb, const float* __restrict__ raw_sigma,
                                    int64_t n_cells, int S, int C, float* __restrict__ out) {
  const int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (idx >= n_cells * (C + 1)) return;
  const int64_t cell = idx / (C + 1);
  const int j = (int)(idx - cell * (C + 1));
  float sum = 0.f;
  for (int s = 0; s < S; ++s) {
    const int64_t row = cell * S + s;
    sum += j < C ? raw_rgb[row * C + j] : raw_sigma[row];
  }
  out[idx] = sum / (float)S;
}

int launch_mean_samples(const float* raw_rgb, const float* raw_sigma, int64_t n_cells, int S, int C, float* out,
                        hipStream_t s) {
  if (n_cells == 0) return PXO_OK;
  const int64_t n = n_cells * (C + 1);
  hipLaunchKernelGGL(mean_samples_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, raw_rgb, raw_sigma,
                     n_cells, S, C, out);
  return check_launch("mean_samples");
}

// ------------------------------------------------------------------------------------------
// Philox4x32-10 uniform generator
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
  const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
  const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
  c[0] = n0; c[1] = (uint32_t)p1; c[2] = n2; c[3] = (uint32_t)p0;
}

struct UniformJobs { UniformJob job[3]; int n_jobs; const float* sq_x; int64_t sq_n; float* sq_partial; unsigned int* zero_words; int n_zero; };

// blockIdx.y selects the job; element idx of a job is word idx % 4 of the Philox block with counter idx / 4 and the
// job's stream id in the upper counter words
__global__ __launch_bounds__(256) void uniform_kernel(uint64_t seed, UniformJobs J) {
  // the tile counters of the step's persistent MLP launches (mlp_kernels.hip TileTicket) start from zero
  if (blockIdx.x == 0 && blockIdx.y == 0 && (int)threadIdx.x < J.n_zero) J.zero_words[threadIdx.x] = 0u;
  if ((int)blockIdx.y == J.n_jobs) {
    if (J.sq_x == nullptr) return;
    // parameter-norm partials (weight_l2 = sum(p^2) / n, train.py:101-108): kSumsqBlocks blocks, strided loads, fixed-order
    // tree in LDS; finalize_stats_kernel adds the partials in order
    if (blockIdx.x >= kSumsqBlocks) return;
    __shared__ float red[256];
    float acc = 0.f;
    for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < J.sq_n; i += (int64_t)kSumsqBlocks * 256)
      acc += J.sq_x[i] * J.sq_x[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
      if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
      __syncthreads();
    }
    if (threadIdx.x == 0) J.sq_partial[blockIdx.x] = red[0];
    return;
  }
  const UniformJob& jb = J.job[blockIdx.y];
  const int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (q * 4 >= jb.n) return;
  uint32_t c[4] = {(uint32_t)q, (uint32_t)(q >> 32), (uint32_t)jb.stream_id, (uint32_t)(jb.stream_id >> 32)};
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t idx = q * 4 + i;
    if (idx < jb.n) {
      const float r01 = (float)(c[i] >> 8) * (1.0f / 16777216.0f);   // [0,1), 24 bits
      jb.out[idx] = jb.lo + (jb.hi - jb.lo) * r01;
    }
  }
}

__global__ void randint_kernel(uint64_t seed, uint64_t stream_id, int64_t count, int64_t n, int64_t* __restrict__ out) {
  const int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (q * 2 >= count) return;
  uint32_t c[4] = {(uint32_t)q, (uint32_t)(q >> 32), (uint32_t)stream_id, (uint32_t)(stream_id >> 32)};
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int64_t idx = q * 2 + i;
    if (idx < count) {
      const uint64_t r64 = ((uint64_t)c[2 * i] << 32) | c[2 * i + 1];
      out[idx] = (int64_t)(r64 % (uint64_t)n);       // bias < n/2^64, negligible
    }
  }
}

int launch_randint(uint64_t seed, uint64_t stream_id, int64_t count, int64_t n, int64_t* out, hipStream_t s) {
  if (count == 0) return PXO_OK;
  const int64_t q = (count + 1) / 2;
  hipLaunchKernelGGL(randint_kernel, dim3((unsigned)((q + 255) / 256)), dim3(256), 0, s, seed, stream_id, count, n, out);
  return check_launch("randint");
}

// Dataset._next_train for one image (nerf_sh/nerf/datasets.py:159-166) in ONE launch: the pixel ids of randint_kernel
// (element i = 64-bit pair i % 2 of Philox block i / 2, mod W*H), the rays of generate_rays_kernel and the gather of the
// image's colours -- the same arithmetic as the three separate launches, bit for bit.
__global__ void sample_batch_kernel(uint64_t seed, uint64_t stream_id, const float* __restrict__ c2w, int W, int H, float focal,
                                    const float* __restrict__ image, int64_t B, int64_t first, int64_t* __restrict__ ids,
                                    float* __restrict__ o, float* __restrict__ d, float* __restrict__ v,
                                    float* __restrict__ pixels) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= B) return;
  const int64_t e = first + i;            // element of the stream: output row i is element first + i of the global draw
  const int64_t q = e >> 1;
  uint32_t c[4] = {(uint32_t)q, (uint32_t)(q >> 32), (uint32_t)stream_id, (uint32_t)(stream_id >> 32)};
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  const int h = (int)(e & 1);
  const uint64_t r64 = ((uint64_t)(h ? c[2] : c[0]) << 32) | (h ? c[3] : c[1]);
  const int64_t p = (int64_t)(r64 % (uint64_t)((int64_t)W * H));
  if (ids) ids[i] = p;
  pixel_ray(c2w, W, H, focal, p, i, o, d, v);
#pragma unroll
  for (int a = 0; a < 3; ++a) pixels[i * 3 + a] = image[p * 3 + a];
}

int launch_sample_batch(uint64_t seed, uint64_t stream_id, const float* c2w, int W, int H, float focal, const float* image,
                        int64_t B, int64_t first, int64_t* ids, float* o, float* d, float* v, float* pixels, hipStream_t s) {
  if (B == 0) return PXO_OK;
  hipLaunchKernelGGL(sample_batch_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, s, seed, stream_id, c2w, W, H, focal,
                     image, B, first, ids, o, d, v, pixels);
  return check_launch("sample_batch");
}

int launch_uniform_jobs(uint64_t seed, const UniformJob* jobs, int n_jobs, hipStream_t s, const float* sq_x, int64_t sq_n,
                        float* sq_partial, unsigned int* zero_words, int n_zero) {
  UniformJobs J;
  int64_t qmax = 0;
  int nj = 0;
  for (int i = 0; i < n_jobs && nj < 3; ++i) {
    if (jobs[i].n <= 0) continue;
    J.job[nj++] = jobs[i];
    const int64_t q = (jobs[i].n + 3) / 4;
    if (q > qmax) qmax = q;
  }
  const bool sq = sq_x != nullptr && sq_partial != nullptr;
  if (zero_words == nullptr || n_zero < 0) n_zero = 0;
  if (n_zero > 256) { set_error("uniform: at most 256 words can be zeroed by the launch (got %d)", n_zero); return PXO_ERR_ARG; }
  if (nj == 0 && !sq && n_zero == 0) return PXO_OK;
  for (int i = nj; i < 3; ++i) J.job[i] = UniformJob{0, 0, 0.f, 0.f, nullptr};
  J.n_jobs = nj; J.sq_x = sq ? sq_x : nullptr; J.sq_n = sq_n; J.sq_partial = sq_partial;
  J.zero_words = zero_words; J.n_zero = n_zero;
  int64_t bx = (qmax + 255) / 256;
  if (sq && bx < kSumsqBlocks) bx = kSumsqBlocks;
  if (bx < 1) bx = 1;
  const int by = nj + (sq ? 1 : 0);
  hipLaunchKernelGGL(uniform_kernel, dim3((unsigned)bx, by > 0 ? by : 1), dim3(256), 0, s, seed, J);
  return check_launch("uniform");
}

// add_gaussian_noise (nerf_sh/nerf/model_utils.py:317-332): raw += noise_std * N(0,1).  Injected draws, or Box-Muller on the
// Philox block of the element's quad (layout in include/plenoctree_hip.h).
__global__ void add_noise_kernel(float* __restrict__ raw, int64_t n, float noise_std, const float* __restrict__ noise,
                                 uint64_t seed, uint64_t stream_id) {
  const int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (q * 4 >= n) return;
  float z[4];
  if (noise) {
#pragma unroll
    for (int i = 0; i < 4; ++i) z[i] = q * 4 + i < n ? noise[q * 4 + i] : 0.0f;
  } else {
    uint32_t c[4] = {(uint32_t)q, (uint32_t)(q >> 32), (uint32_t)stream_id, (uint32_t)(stream_id >> 32)};
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
      philox_round(c, k0, k1);
      k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const float u1 = (float)((c[2 * h] >> 8) + 1u) * (1.0f / 16777216.0f);      // (0, 1]
      const float u2 = (float)(c[2 * h + 1] >> 8) * (1.0f / 16777216.0f);         // [0, 1)
      const float r = sqrtf(-2.0f * logf(u1));
      float sn, cs;
      sincospif(2.0f * u2, &sn, &cs);
      z[2 * h] = r * cs;
      z[2 * h + 1] = r * sn;
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t idx = q * 4 + i;
    if (idx < n) raw[idx] = raw[idx] + noise_std * z[i];
  }
}

int launch_add_noise(float* raw, int64_t n, float noise_std, const float* noise, uint64_t seed, uint64_t stream_id, hipStream_t s) {
  if (n == 0) return PXO_OK;
  const int64_t q = (n + 3) / 4;
  hipLaunchKernelGGL(add_noise_kernel, dim3((unsigned)((q + 255) / 256)), dim3(256), 0, s, raw, n, noise_std, noise, seed, stream_id);
  return check_launch("add_gaussian_noise");
}

int launch_uniform(uint64_t seed, uint64_t stream_id, int64_t n, float lo, float hi, float* out, hipStream_t s) {
  const UniformJob job{stream_id, n, lo, hi, out};
  return launch_uniform_jobs(seed, &job, 1, s);
}

}  // namespace pxo
