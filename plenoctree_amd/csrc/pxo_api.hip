// C ABI of the gfx950 NeRF-SH hot path (see include/plenoctree_hip.h).  Host-side glue only:
// argument checks, workspace carving and the kernel sequences of NerfModel.__call__
// (nerf_sh/nerf/models.py:216-348) and loss_fn/value_and_grad (nerf_sh/train.py:66-116).
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "pxo_common.h"

namespace pxo {

static thread_local std::string g_last_error;

void set_error(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return PXO_ERR_HIP;
  }
  return PXO_OK;
}

int num_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
    if (n <= 0) n = 256;
    if (2 * n > kMaxMlpGrid) n = kMaxMlpGrid / 2;
  }
  return n;
}

// run-time choices between implementations of the same result (pxo_set_tuning)
// tile schedule: the device counter by default (r05b: 3.548 vs 3.556 ms per step at 512 rays, 23.78 vs 23.84 at 4096 in one
// process, profiles/r05b_tune_ab.jsonl; and a workgroup that starts late is not the launch's tail, r05b_contention_probe.jsonl)
// (atomics: a knob may be set from one host thread while another enqueues a step; each step reads a knob at most once per
// decision and records what it decided, see g_step_skipped below)
static std::atomic<int> g_tune_tile_sched{1};
static std::atomic<int> g_tune_wgrad_ranges{0};
static std::atomic<int> g_tune_wgrad_skinny_ranges{0};
static std::atomic<int> g_tune_x6_wgrad{1};
// PXO_TUNE_COARSE_REVERSE_STREAM: 1 = the reverse pass of the coarse level (backward(data), weight gradients, slab reduce of
// MLP_0) runs on an internal side stream beside the fine level's forward, whose ragged last round of tiles leaves CUs idle
// (3.3 rounds at 512 rays per GPU); 0 (default) = everything on the caller's stream.  Same kernels, same sums: bits unchanged.
// Measured (r06b): 3.5905 / 3.5849 vs 3.5909 / 3.5969 ms per 512-ray step, 23.56 vs 23.72 ms at 4096 rays -- +0.2 % / +0.7 %,
// and per-kernel HIP-event times stop meaning anything (the fine forward shares the GPU with the coarse reverse: 4.83 ms
// "per launch" instead of 3.95), so the measured default stays the single stream.
static std::atomic<int> g_tune_coarse_stream{0};
static hipStream_t g_side_stream = nullptr;           // created on first use, lives for the process
static hipEvent_t g_ev_fork = nullptr, g_ev_join = nullptr;
static std::mutex g_side_mu;
static bool side_stream_ready() {
  std::lock_guard<std::mutex> lk(g_side_mu);
  if (g_side_stream) return true;
  int lo = 0, hi = 0;
  (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
  if (hipStreamCreateWithPriority(&g_side_stream, hipStreamNonBlocking, lo) != hipSuccess) { g_side_stream = nullptr; return false; }
  if (hipEventCreateWithFlags(&g_ev_fork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&g_ev_join, hipEventDisableTiming) != hipSuccess) {
    (void)hipStreamDestroy(g_side_stream);
    g_side_stream = nullptr;
    return false;
  }
  return true;
}
// Did the last pxo_train_fwd_bwd* call on a workspace run its reverse pass in skipping mode?  Decided once per step from the
// cfg AND the split-K tuning in force at that moment; pxo_train_backward_work reports for THAT decision, whatever
// pxo_set_tuning did since.
static std::mutex g_step_mu;
static std::unordered_map<const void*, bool> g_step_skipped;
int tune_tile_sched() { return g_tune_tile_sched; }
int tune_wgrad_ranges() { return g_tune_wgrad_ranges; }
int tune_wgrad_skinny_ranges() { return g_tune_wgrad_skinny_ranges; }
int tune_x6_wgrad() { return g_tune_x6_wgrad; }

int validate_cfg(const PxoCfg* cfg) {
  PXO_REQUIRE(cfg != nullptr, "cfg is NULL");
  PXO_REQUIRE(cfg->sh_deg >= 0 && cfg->sh_deg <= 4, "sh_deg %d not in [0,4] (nerf_sh/nerf/sh.py:69)", cfg->sh_deg);
  PXO_REQUIRE(cfg->min_deg_point == 0 && cfg->max_deg_point == 10,
              "only min_deg_point=0,max_deg_point=10 is built (got %d,%d)", cfg->min_deg_point, cfg->max_deg_point);
  PXO_REQUIRE(cfg->num_coarse_samples >= 3 && cfg->num_coarse_samples <= 128, "num_coarse_samples %d not in [3,128]",
              cfg->num_coarse_samples);
  PXO_REQUIRE(cfg->num_fine_samples >= 0 && cfg->num_coarse_samples + cfg->num_fine_samples <= 256,
              "num_coarse_samples+num_fine_samples must be <= 256");
  PXO_REQUIRE(cfg->mlp_precision == PXO_MLP_F32 || cfg->mlp_precision == PXO_MLP_BF16X3 || cfg->mlp_precision == PXO_MLP_BF16X6,
              "mlp_precision %d unknown", cfg->mlp_precision);
  PXO_REQUIRE(cfg->noise_std >= 0.f, "noise_std %g < 0 (0 = None)", (double)cfg->noise_std);
  PXO_REQUIRE(cfg->skip_zero_rows == 0 || cfg->skip_zero_rows == 1, "skip_zero_rows %d is not 0 / 1", cfg->skip_zero_rows);
  return PXO_OK;
}

// ---- HIP-event profiler ------------------------------------------------------------------
struct ProfRecord { hipEvent_t a, b; int tag; int64_t rows; };
static unsigned g_prof_mask = 0;              // bit t set: launches tagged t are bracketed by events
static std::vector<ProfRecord> g_prof;

KernelTimer::KernelTimer(int tag, int64_t rows, hipStream_t s) : slot(-1), stream(s) {
  if (!((g_prof_mask >> tag) & 1u)) return;
  ProfRecord r;
  if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
  r.tag = tag; r.rows = rows;
  (void)hipEventRecord(r.a, s);
  g_prof.push_back(r);
  slot = (int)g_prof.size() - 1;
}
KernelTimer::~KernelTimer() {
  if (slot >= 0) (void)hipEventRecord(g_prof[slot].b, stream);
}

// bump allocator over the caller's workspace; with base == nullptr it only measures
struct Carver {
  char* base;
  size_t off = 0;
  explicit Carver(void* b) : base(reinterpret_cast<char*>(b)) {}
  template <typename T>
  T* take(int64_t count) {
    off = (off + 255) & ~(size_t)255;
    T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off += (size_t)count * sizeof(T);
    return p;
  }
};

struct PassBuffers {      // one MLP pass (coarse or fine)
  int64_t M = 0;          // rows through the MLP (ray samples + appended sparsity points)
  int S = 0;              // samples per ray
  float *z, *pts, *raw_rgb, *raw_sigma, *acts, *enc, *comp_rgb, *disp, *acc, *weights;
  uint32_t* mask;
  float *ray_sse, *d_raw_rgb, *d_raw_sigma, *dz, *dbias;
  uint8_t* live;          // one flag per kLiveRows rows: written by the backward(data) kernel, read by the wgrad kernels
};

static void carve_pass(Carver& c, PassBuffers& p, int64_t B, int S, int64_t extra_rows, int C, bool train,
                       bool own_outputs) {
  p.S = S;
  p.M = B * S + extra_rows;
  p.z = c.take<float>(B * S);
  p.pts = c.take<float>(p.M * 3);
  p.raw_rgb = c.take<float>(p.M * C);
  p.raw_sigma = c.take<float>(p.M);
  p.weights = c.take<float>(B * S);
  if (own_outputs) {
    p.comp_rgb = c.take<float>(B * 3);
    p.disp = c.take<float>(B);
    p.acc = c.take<float>(B);
  }
  if (train) {
    p.acts = c.take<float>(p.M * kW * kDepth);
    p.enc = c.take<float>(p.M * kEncPad);
    p.mask = c.take<uint32_t>(mask_words(p.M));
    p.ray_sse = c.take<float>(B);
    p.d_raw_rgb = c.take<float>(p.M * C);
    p.d_raw_sigma = c.take<float>(p.M);
    p.dz = c.take<float>(p.M * kW * kDepth);
    p.dbias = c.take<float>(dbias_floats(p.M));
    p.live = c.take<uint8_t>(live_flags(p.M));
  } else {
    p.acts = p.enc = p.ray_sse = p.d_raw_rgb = p.d_raw_sigma = p.dz = p.dbias = nullptr;
    p.mask = nullptr;
    p.live = nullptr;
  }
}

struct TrainWs {
  PassBuffers c, f;
  float *t_rand, *u, *sp, *scalars, *sp_exp;
  void* wgrad_ws;
  size_t wgrad_bytes;
  int64_t n_sp;
  size_t total;
};

static void carve_train(const PxoCfg* cfg, int64_t B, void* ws, bool train, TrainWs& t) {
  Carver c(ws);
  const int C = rgb_channels(cfg->sh_deg);
  const int Nc = cfg->num_coarse_samples, Nf = cfg->num_fine_samples;
  const bool sp = train && cfg->sparsity_weight > 0.f && cfg->sparsity_npoints > 0;
  t.n_sp = sp ? cfg->sparsity_npoints : 0;
  // eval_points_raw uses MLP_1 when there is a fine pass (nerf_sh/nerf/models.py:165-168), so the
  // sparsity points ride along as extra rows of the last pass.
  carve_pass(c, t.c, B, Nc, Nf > 0 ? 0 : t.n_sp, C, train, train);
  if (Nf > 0) carve_pass(c, t.f, B, Nc + Nf, t.n_sp, C, train, train);
  t.t_rand = c.take<float>(B * Nc);
  t.u = c.take<float>(B * (Nf > 0 ? Nf : 1));
  t.sp = c.take<float>(t.n_sp * 3 + 4);
  t.scalars = c.take<float>(128);
  t.sp_exp = c.take<float>(t.n_sp + 4);
  if (train) {
    const int64_t Mmax = Nf > 0 ? t.f.M : t.c.M;
    t.wgrad_bytes = wgrad_workspace_bytes(cfg, Mmax);
    t.wgrad_ws = c.take<char>((int64_t)t.wgrad_bytes);
  } else {
    t.wgrad_bytes = 0;
    t.wgrad_ws = nullptr;
  }
  t.total = (c.off + 255) & ~(size_t)255;
}

#define PXO_TRY(expr)            \
  do {                           \
    int _rc = (expr);            \
    if (_rc != PXO_OK) return _rc; \
  } while (0)

// NerfModel.__call__ (nerf_sh/nerf/models.py:216-348) in three pieces, so that the training sequence can put the reverse of
// the coarse level between the levels.  pixels != nullptr selects the training form: compositing, the pixel loss and its
// reverse in one kernel per pass (d_raw_* and ray_sse are written, the rgb/disp/acc outputs are not), with the sparsity
// rows of the last pass served by the same launch.
struct Draws { const float* t_rand; const float* u; bool noisy; uint64_t seed; };

// every uniform draw of the step in one launch (jax.random.uniform call sites model_utils.py:135,262, train.py:79)
static int prepare_draws(const PxoCfg* cfg, TrainWs& t, int64_t B, int randomized, const float* t_rand, const float* u,
                         const float* sp_points, uint64_t seed, hipStream_t s, Draws& out, const float* params = nullptr,
                         int64_t n_params = 0, float* sumsq_partial = nullptr, unsigned int* zero_words = nullptr,
                         int n_zero = 0) {
  const int Nc = cfg->num_coarse_samples, Nf = cfg->num_fine_samples;
  PassBuffers& last = Nf > 0 ? t.f : t.c;
  UniformJob jobs[3];
  int nj = 0;
  if (randomized && !t_rand) { jobs[nj++] = UniformJob{0, B * Nc, 0.f, 1.f, t.t_rand}; t_rand = t.t_rand; }
  if (randomized && Nf > 0 && !u) { jobs[nj++] = UniformJob{1, B * Nf, 0.f, 1.f, t.u}; u = t.u; }
  if (!randomized) { t_rand = nullptr; u = nullptr; }
  if (t.n_sp > 0) {
    float* dst = last.pts + B * last.S * 3;
    if (sp_points) {
      if (hipMemcpyAsync(dst, sp_points, (size_t)t.n_sp * 3 * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess) {
        set_error("hipMemcpyAsync(sp_points) failed");
        return PXO_ERR_HIP;
      }
    } else {
      jobs[nj++] = UniformJob{2, t.n_sp * 3, -cfg->sparsity_radius, cfg->sparsity_radius, dst};
    }
  }
  out.t_rand = t_rand; out.u = u;
  out.noisy = randomized != 0 && cfg->noise_std > 0.f;     // (noise_std is not None) and randomized, model_utils.py:329
  out.seed = seed;
  return launch_uniform_jobs(seed, jobs, nj, s, params, n_params, sumsq_partial, zero_words, n_zero);
}

static int forward_coarse(const PxoCfg* cfg, TrainWs& t, const float* pk0, const float* o, const float* d, const float* v,
                          int64_t B, const Draws& dr, const float* pixels, float* rgb_c, float* disp_c, float* acc_c,
                          hipStream_t s, unsigned int* tile_counter = nullptr) {
  const int Nc = cfg->num_coarse_samples, Nf = cfg->num_fine_samples;
  PXO_TRY(launch_sample_along_rays(o, d, B, Nc, cfg->near_, cfg->far_, cfg->lindisp, dr.t_rand, t.c.z, t.c.pts, s));
  PXO_TRY(launch_mlp_fwd(cfg, pk0, t.c.pts, t.c.M, t.c.raw_rgb, t.c.raw_sigma, t.c.acts, t.c.enc, t.c.mask, s, tile_counter));
  if (dr.noisy) PXO_TRY(launch_add_noise(t.c.raw_sigma, B * Nc, cfg->noise_std, nullptr, dr.seed, 3, s));   // models.py:258-264
  if (pixels)
    return launch_shade_composite_train(cfg, t.c.raw_rgb, t.c.raw_sigma, t.c.z, d, v, pixels, B, Nc, nullptr,
                                        Nf > 0 ? t.c.weights : nullptr, t.c.ray_sse, t.c.d_raw_rgb, t.c.d_raw_sigma,
                                        Nf > 0 ? 0 : t.n_sp, t.sp_exp, s);
  return launch_shade_composite_fwd(cfg, t.c.raw_rgb, t.c.raw_sigma, t.c.z, d, v, B, Nc, rgb_c, disp_c, acc_c,
                                    t.c.weights, s);
}

static int forward_fine(const PxoCfg* cfg, TrainWs& t, const float* pk1, const float* o, const float* d, const float* v,
                        int64_t B, const Draws& dr, const float* pixels, float* rgb_f, float* disp_f, float* acc_f,
                        hipStream_t s, unsigned int* tile_counter = nullptr) {
  const int Nc = cfg->num_coarse_samples, Nf = cfg->num_fine_samples;
  PXO_TRY(launch_sample_pdf(t.c.z, t.c.weights, o, d, B, Nc, Nf, dr.u, t.f.z, t.f.pts, s));
  PXO_TRY(launch_mlp_fwd(cfg, pk1, t.f.pts, t.f.M, t.f.raw_rgb, t.f.raw_sigma, t.f.acts, t.f.enc, t.f.mask, s, tile_counter));
  if (dr.noisy) PXO_TRY(launch_add_noise(t.f.raw_sigma, B * (Nc + Nf), cfg->noise_std, nullptr, dr.seed, 4, s));   // :318-324
  if (pixels)
    return launch_shade_composite_train(cfg, t.f.raw_rgb, t.f.raw_sigma, t.f.z, d, v, pixels, B, Nc + Nf, nullptr,
                                        nullptr, t.f.ray_sse, t.f.d_raw_rgb, t.f.d_raw_sigma, t.n_sp, t.sp_exp, s);
  return launch_shade_composite_fwd(cfg, t.f.raw_rgb, t.f.raw_sigma, t.f.z, d, v, B, Nc + Nf, rgb_f, disp_f, acc_f,
                                    t.f.weights, s);
}

// diagnostic: `blocks` workgroups of `threads` threads that do nothing for `micros` microseconds of the device's
// constant-rate clock (what a ring all-reduce's kernel looks like to the kernels it shares the GPU with)
__global__ void occupy_kernel(long long ticks) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

}  // namespace pxo

using namespace pxo;

extern "C" {

const char* pxo_last_error(void) { return g_last_error.c_str(); }
int pxo_version(void) { return PXO_ABI_VERSION; }
size_t pxo_cfg_bytes(void) { return sizeof(PxoCfg); }

int pxo_set_tuning(int knob, int value) {
  switch (knob) {
    case PXO_TUNE_TILE_SCHED:
      PXO_REQUIRE(value == 0 || value == 1, "pxo_set_tuning: tile schedule must be 0 (static stride) or 1 (device counter), got %d", value);
      g_tune_tile_sched = value;
      return PXO_OK;
    case PXO_TUNE_WGRAD_RANGES:
      PXO_REQUIRE(value >= 0 && value <= num_cus(), "pxo_set_tuning: row ranges per layer must be 0 (built-in choice) .. %d, got %d",
                  num_cus(), value);
      g_tune_wgrad_ranges = value;
      return PXO_OK;
    case PXO_TUNE_COARSE_REVERSE_STREAM:
      PXO_REQUIRE(value == 0 || value == 1, "pxo_set_tuning: coarse reverse stream must be 0 (caller's stream) or 1 (side stream), got %d", value);
      g_tune_coarse_stream = value;
      return PXO_OK;
    case PXO_TUNE_WGRAD_SKINNY_RANGES:
      PXO_REQUIRE(value >= 0 && value <= 2 * num_cus(), "pxo_set_tuning: row ranges of the skinny products must be 0 (built-in choice) .. %d, got %d",
                  2 * num_cus(), value);
      g_tune_wgrad_skinny_ranges = value;
      return PXO_OK;
    case PXO_TUNE_X6_WGRAD:
      PXO_REQUIRE(value == 0 || value == 1, "pxo_set_tuning: bf16x6 weight gradients must be 0 (float32 MFMA) or 1 (bf16x6), got %d", value);
      g_tune_x6_wgrad = value;
      return PXO_OK;
    default:
      set_error("pxo_set_tuning: unknown knob %d", knob);
      return PXO_ERR_ARG;
  }
}
int pxo_get_tuning(int knob, int* value) {
  PXO_REQUIRE(value != nullptr, "pxo_get_tuning: NULL pointer");
  switch (knob) {
    case PXO_TUNE_TILE_SCHED: *value = g_tune_tile_sched; return PXO_OK;
    case PXO_TUNE_WGRAD_RANGES: *value = g_tune_wgrad_ranges; return PXO_OK;
    case PXO_TUNE_WGRAD_SKINNY_RANGES: *value = g_tune_wgrad_skinny_ranges; return PXO_OK;
    case PXO_TUNE_COARSE_REVERSE_STREAM: *value = g_tune_coarse_stream; return PXO_OK;
    case PXO_TUNE_X6_WGRAD: *value = g_tune_x6_wgrad; return PXO_OK;
    default: set_error("pxo_get_tuning: unknown knob %d", knob); return PXO_ERR_ARG;
  }
}
int pxo_tile_rows(void) { return kTM; }

int pxo_param_layout(const PxoCfg* cfg, PxoLeaf* leaves, int64_t* floats_per_mlp) {
  PXO_TRY(validate_cfg(cfg));
  const int deg = cfg->sh_deg;
  if (leaves) {
    for (int l = 0; l < 10; ++l) {
      leaves[2 * l] = PxoLeaf{l, 0, leaf_kernel_off(l, deg), layer_in(l), layer_out(l, deg)};
      leaves[2 * l + 1] = PxoLeaf{l, 1, leaf_bias_off(l, deg), layer_out(l, deg), 1};
    }
  }
  if (floats_per_mlp) *floats_per_mlp = mlp_param_count(deg);
  return PXO_OK;
}

int pxo_packed_sizes(const PxoCfg* cfg, int64_t* fwd_floats, int64_t* bwd_floats) {
  PXO_TRY(validate_cfg(cfg));
  const bool x6 = cfg->mlp_precision == PXO_MLP_BF16X6;
  if (fwd_floats) *fwd_floats = x6 ? x6_fwd_image_floats(cfg->sh_deg) : fwd_image_floats(cfg->sh_deg);
  if (bwd_floats) *bwd_floats = x6 ? x6_bwd_image_floats(cfg->sh_deg) : bwd_image_floats(cfg->sh_deg);
  return PXO_OK;
}

int pxo_pack_weights(const PxoCfg* cfg, const float* mlp_params, float* packed_fwd, float* packed_bwd, void* stream) {
  PXO_TRY(validate_cfg(cfg));
  PXO_REQUIRE(mlp_params && packed_fwd, "pxo_pack_weights: NULL pointer");
  return launch_pack(cfg, mlp_params, packed_fwd, packed_bwd, (hipStream_t)stream);
}

int pxo_sample_along_rays(const float* origins, const float* directions, int64_t B, int S, float near_, float far_,
                          int lindisp, const float* t_rand, float* z_vals, float* pts, void* stream) {
  PXO_REQUIRE(B >= 0 && S >= 1, "pxo_sample_along_rays: bad sizes B=%lld S=%d", (long long)B, S);
  if (B == 0) return PXO_OK;                   // empty input: nothing to check, nothing to launch
  PXO_REQUIRE(origins && directions && z_vals && pts, "pxo_sample_along_rays: NULL pointer");
  return launch_sample_along_rays(origins, directions, B, S, near_, far_, lindisp, t_rand, z_vals, pts,
                                  (hipStream_t)stream);
}

int pxo_posenc(const float* x, int64_t N, float* enc, void* stream) {
  if (N == 0) return PXO_OK;
  PXO_REQUIRE(N >= 0 && x && enc, "pxo_posenc: bad arguments");
  return launch_posenc(x, N, enc, (hipStream_t)stream);
}

size_t pxo_relu_mask_bytes(int64_t M) { return (size_t)mask_words(M) * sizeof(uint32_t); }
size_t pxo_dbias_partial_bytes(int64_t M) { return (size_t)dbias_floats(M) * sizeof(float); }

int pxo_mlp_fwd(const PxoCfg* cfg, const float* packed_fwd, const float* pts, int64_t M, float* raw_rgb,
                float* raw_sigma, float* acts, float* enc, void* relu_mask, void* stream) {
  PXO_TRY(validate_cfg(cfg));
  if (M == 0) return PXO_OK;
  PXO_REQUIRE(M >= 0 && packed_fwd && pts && raw_sigma, "pxo_mlp_fwd: bad arguments");
  PXO_REQUIRE((acts != nullptr) == (enc != nullptr) && (acts != nullptr) == (relu_mask != nullptr),
              "pxo_mlp_fwd: acts, enc and relu_mask must be all set or all NULL");
  return launch_mlp_fwd(cfg, packed_fwd, pts, M, raw_rgb, raw_sigma, acts, enc, (uint32_t*)relu_mask,
                        (hipStream_t)stream);
}

int pxo_mlp_bwd_data(const PxoCfg* cfg, const float* packed_bwd, const float* d_raw_rgb, const float* d_raw_sigma,
                     const void* relu_mask, int64_t M, float* dz, float* dbias_partial, void* stream) {
  PXO_TRY(validate_cfg(cfg));
  if (M == 0) return PXO_OK;
  PXO_REQUIRE(M >= 0 && packed_bwd && d_raw_rgb && d_raw_sigma && relu_mask && dz && dbias_partial,
              "pxo_mlp_bwd_data: bad arguments");
  return launch_mlp_bwd_data(cfg, packed_bwd, d_raw_rgb, d_raw_sigma, (const uint32_t*)relu_mask, M, dz,
                             dbias_partial, nullptr, nullptr, (hipStream_t)stream);
}

int pxo_wgrad_workspace_bytes(const PxoCfg* cfg, int64_t M, size_t* bytes) {
  PXO_TRY(validate_cfg(cfg));
  PXO_REQUIRE(bytes && M >= 0, "pxo_wgrad_workspace_bytes: bad arguments");
  *bytes = wgrad_workspace_bytes(cfg, M > 0 ? M : 1);
  return PXO_OK;
}

int pxo_mlp_bwd_weights(const PxoCfg* cfg, const float* acts, const float* enc, const float* dz,
                        const float* d_raw_rgb, const float* d_raw_sigma, const float* dbias_partial, int64_t M,
                        float* grads, void* ws, size_t ws_bytes, void* stream) {
  PXO_TRY(validate_cfg(cfg));
  PXO_REQUIRE(M >= 0 && acts && enc && dz && d_raw_rgb && d_raw_sigma && dbias_partial && grads && ws,
              "pxo_mlp_bwd_weights: bad arguments");
  return launch_mlp_bwd_weights(cfg, acts, enc, dz, d_raw_rgb, d_raw_sigma, dbias_partial, M, grads, ws, ws_bytes,
                                nullptr, (hipStream_t)stream);
}

int pxo_shade_composite_fwd(const PxoCfg* cfg, const float* raw_rgb, const float* raw_sigma, const float* z_vals,
                            const float* directions, const float* viewdirs, int64_t B, int S, float* comp_rgb,
                            float* disp, float* acc, float* weights, void* stream) {
  PXO_TRY(validate_cfg(cfg));
  if (B == 0) return PXO_OK;
  PXO_REQUIRE(B >= 0 && raw_rgb && raw_sigma && z_vals && directions && viewdirs && comp_rgb && disp && acc && weights,
              "pxo_shade_composite_fwd: bad arguments");
  return launch_shade_composite_fwd(cfg, raw_rgb, raw_sigma, z_vals, directions, viewdirs, B, S, comp_rgb, disp, acc,
                                    weights, (hipStream_t)stream);
}

int pxo_shade_composite_bwd(const PxoCfg* cfg, const float* raw_rgb, const float* raw_sigma, const float* z_vals,
                            const float* directions, const float* viewdirs, const float* d_comp_rgb, int64_t B, int S,
                            float* d_raw_rgb, float* d_raw_sigma, void* stream) {
  PXO_TRY(validate_cfg(cfg));
  if (B == 0) return PXO_OK;
  PXO_REQUIRE(B >= 0 && raw_rgb && raw_sigma && z_vals && directions && viewdirs && d_comp_rgb && d_raw_rgb &&
                  d_raw_sigma,
              "pxo_shade_composite_bwd: bad arguments");
  return launch_shade_composite_bwd(cfg, raw_rgb, raw_sigma, z_vals, directions, viewdirs, d_comp_rgb, B, S,
                                    d_raw_rgb, d_raw_sigma, (hipStream_t)stream);
}

int pxo_shade_composite_train(const PxoCfg* cfg, const float* raw_rgb, const float* raw_sigma, const float* z_vals,
                              const float* directions, const float* viewdirs, const float* pixels, int64_t B, int S,
                              float* comp_rgb, float* weights, float* ray_sse, float* d_raw_rgb, float* d_raw_sigma,
                              int64_t n_sp, float* sp_exp, void* stream) {
  PXO_TRY(validate_cfg(cfg));
  if (B == 0) return PXO_OK;
  PXO_REQUIRE(B >= 0 && n_sp >= 0 && raw_rgb && raw_sigma && z_vals && directions && viewdirs && pixels && ray_sse &&
                  d_raw_rgb && d_raw_sigma && (n_sp == 0 || sp_exp),
              "pxo_shade_composite_train: bad arguments");
  return launch_shade_composite_train(cfg, raw_rgb, raw_sigma, z_vals, directions, viewdirs, pixels, B, S, comp_rgb,
                                      weights, ray_sse, d_raw_rgb, d_raw_sigma, n_sp, sp_exp, (hipStream_t)stream);
}

int pxo_sample_pdf(const float* z_coarse, const float* w_coarse, const float* origins, const float* directions,
                   int64_t B, int Nc, int Nf, const float* u, float* z_out, float* pts, void* stream) {
  if (B == 0) return PXO_OK;
  PXO_REQUIRE(B >= 0 && z_coarse && w_coarse && origins && directions && z_out && pts, "pxo_sample_pdf: bad arguments");
  return launch_sample_pdf(z_coarse, w_coarse, origins, directions, B, Nc, Nf, u, z_out, pts, (hipStream_t)stream);
}

int pxo_uniform(uint64_t seed, uint64_t stream_id, int64_t n, float lo, float hi, float* out, void* stream) {
  if (n == 0) return PXO_OK;
  PXO_REQUIRE(n >= 0 && out, "pxo_uniform: bad arguments");
  return launch_uniform(seed, stream_id, n, lo, hi, out, (hipStream_t)stream);
}

int pxo_add_gaussian_noise(float* raw, int64_t n, float noise_std, const float* noise, uint64_t seed, uint64_t stream_id,
                           void* stream) {
  if (n == 0) return PXO_OK;
  PXO_REQUIRE(n >= 0 && raw && noise_std >= 0.f, "pxo_add_gaussian_noise: bad arguments");
  return launch_add_noise(raw, n, noise_std, noise, seed, stream_id, (hipStream_t)stream);
}

int pxo_randint(uint64_t seed, uint64_t stream_id, int64_t count, int64_t n, int64_t* out, void* stream) {
  if (count == 0) return PXO_OK;
  PXO_REQUIRE(count >= 0 && n >= 1 && out, "pxo_randint: bad arguments");
  return launch_randint(seed, stream_id, count, n, out, (hipStream_t)stream);
}

int pxo_generate_rays(const float* c2w, int W, int H, float focal, const int64_t* pixel_ids, int64_t B,
                      float* origins, float* directions, float* viewdirs, void* stream) {
  if (B == 0) return PXO_OK;
  PXO_REQUIRE(B >= 0 && W >= 1 && H >= 1 && focal > 0.f && c2w && origins && directions && viewdirs,
              "pxo_generate_rays: bad arguments");
  return launch_generate_rays(c2w, 1, W, H, focal, pixel_ids, B, origins, directions, viewdirs, (hipStream_t)stream);
}

int pxo_sample_batch(uint64_t seed, uint64_t stream_id, const float* c2w, int W, int H, float focal, const float* image_rgb,
                     int64_t B, int64_t first, int64_t* pixel_ids, float* origins, float* directions, float* viewdirs,
                     float* pixels, void* stream) {
  if (B == 0) return PXO_OK;
  PXO_REQUIRE(B >= 0 && first >= 0 && W >= 1 && H >= 1 && focal > 0.f && c2w && image_rgb && origins && directions &&
                  viewdirs && pixels,
              "pxo_sample_batch: bad arguments");
  return launch_sample_batch(seed, stream_id, c2w, W, H, focal, image_rgb, B, first, pixel_ids, origins, directions, viewdirs,
                             pixels, (hipStream_t)stream);
}

int pxo_generate_rays_multi(const float* c2w, int n_cams, int W, int H, float focal, const int64_t* ray_ids, int64_t B,
                            float* origins, float* directions, float* viewdirs, void* stream) {
  if (B == 0) return PXO_OK;
  PXO_REQUIRE(B >= 0 && n_cams >= 1 && W >= 1 && H >= 1 && focal > 0.f && c2w && ray_ids && origins && directions &&
                  viewdirs,
              "pxo_generate_rays_multi: bad arguments");
  return launch_generate_rays(c2w, n_cams, W, H, focal, ray_ids, B, origins, directions, viewdirs,
                              (hipStream_t)stream);
}

int pxo_mean_over_samples(const PxoCfg* cfg, const float* raw_rgb, const float* raw_sigma, int64_t n_cells, int S,
                          float* out, void* stream) {
  PXO_TRY(validate_cfg(cfg));
  if (n_cells == 0) return PXO_OK;
  PXO_REQUIRE(n_cells >= 0 && S >= 1 && raw_rgb && raw_sigma && out, "pxo_mean_over_samples: bad arguments");
  return launch_mean_samples(raw_rgb, raw_sigma, n_cells, S, rgb_channels(cfg->sh_deg), out, (hipStream_t)stream);
}

int pxo_adam_step(float* params, float* m, float* v, const float* grads, int64_t n, float lr, int64_t step,
                  float grad_scale, void* stream) {
  if (n == 0) return PXO_OK;
  PXO_REQUIRE(n >= 0 && params && m && v && grads && step >= 0, "pxo_adam_step: bad arguments");
  return launch_adam(params, m, v, grads, n, lr, step, grad_scale, (hipStream_t)stream);
}

int pxo_render_workspace_bytes(const PxoCfg* cfg, int64_t B, size_t* bytes) {
  PXO_TRY(validate_cfg(cfg));
  PXO_REQUIRE(bytes && B >= 0, "pxo_render_workspace_bytes: bad arguments");
  TrainWs t;
  carve_train(cfg, B, nullptr, false, t);
  *bytes = t.total;
  return PXO_OK;
}

int pxo_render_fwd(const PxoCfg* cfg, const float* packed_fwd0, const float* packed_fwd1, const float* origins,
                   const float* directions, const float* viewdirs, int64_t B, int randomized, const float* t_rand,
                   const float* u, uint64_t seed, float* rgb_c, float* disp_c, float* acc_c, float* rgb_f,
                   float* disp_f, float* acc_f, void* ws, size_t ws_bytes, void* stream) {
  PXO_TRY(validate_cfg(cfg));
  if (B == 0) return PXO_OK;                   // an empty batch has no buffers to check
  PXO_REQUIRE(B >= 0 && packed_fwd0 && origins && directions && viewdirs && rgb_c && disp_c && acc_c && ws,
              "pxo_render_fwd: bad arguments");
  if (cfg->num_fine_samples > 0)
    PXO_REQUIRE(packed_fwd1 && rgb_f && disp_f && acc_f, "pxo_render_fwd: fine outputs/weights missing");
  TrainWs t;
  carve_train(cfg, B, ws, false, t);
  if (ws_bytes < t.total) {
    set_error("pxo_render_fwd: workspace %zu < %zu", ws_bytes, t.total);
    return PXO_ERR_WORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  Draws dr;
  PXO_TRY(prepare_draws(cfg, t, B, randomized, t_rand, u, nullptr, seed, s, dr));
  PXO_TRY(forward_coarse(cfg, t, packed_fwd0, origins, directions, viewdirs, B, dr, nullptr, rgb_c, disp_c, acc_c, s));
  if (cfg->num_fine_samples > 0)
    PXO_TRY(forward_fine(cfg, t, packed_fwd1, origins, directions, viewdirs, B, dr, nullptr, rgb_f, disp_f, acc_f, s));
  return PXO_OK;
}

int pxo_train_workspace_bytes(const PxoCfg* cfg, int64_t B, size_t* bytes) {
  PXO_TRY(validate_cfg(cfg));
  PXO_REQUIRE(bytes && B >= 1, "pxo_train_workspace_bytes: bad arguments");
  TrainWs t;
  carve_train(cfg, B, nullptr, true, t);
  *bytes = t.total;
  return PXO_OK;
}

int pxo_train_fwd_bwd_bucketed(const PxoCfg* cfg, const float* params, const float* packed_fwd0, const float* packed_bwd0,
                               const float* packed_fwd1, const float* packed_bwd1, const float* origins,
                               const float* directions, const float* viewdirs, const float* pixels, int64_t B,
                               int randomized, const float* t_rand, const float* u, const float* sp_points, uint64_t seed,
                               float* grads, float* stats, void* ws, size_t ws_bytes, void* grads0_ready, void* stream) {
  PXO_TRY(validate_cfg(cfg));
  PXO_REQUIRE(B >= 1 && params && packed_fwd0 && packed_bwd0 && origins && directions && viewdirs && pixels && grads &&
                  stats && ws,
              "pxo_train_fwd_bwd: bad arguments");
  if (cfg->mlp_precision != PXO_MLP_F32 && cfg->mlp_precision != PXO_MLP_BF16X6) {
    set_error("pxo_train_fwd_bwd: training runs in float32 or its float32-accurate bf16x6 emulation only (mlp_precision bf16x3 is an inference option)");
    return PXO_ERR_UNSUPPORTED;
  }
  if (cfg->num_fine_samples > 0) PXO_REQUIRE(packed_fwd1 && packed_bwd1, "pxo_train_fwd_bwd: MLP_1 images missing");
  hipStream_t s = (hipStream_t)stream;
  TrainWs t;
  carve_train(cfg, B, ws, true, t);
  if (ws_bytes < t.total) {
    set_error("pxo_train_fwd_bwd: workspace %zu < %zu", ws_bytes, t.total);
    return PXO_ERR_WORKSPACE;
  }
  const int deg = cfg->sh_deg;
  const int64_t n_mlp = mlp_param_count(deg);
  const float wd_coef = 2.f * cfg->weight_decay_mult / (float)(2 * n_mlp);   // d/dp of weight_decay_mult * sum(p^2)/n (train.py:101-114)
  // weight_l2 = sum(p^2) / n (train.py:101-108) depends on the parameters only: its partial sums ride in the step's first
  // launch, together with every uniform draw
  float* const sumsq_partial = t.scalars;
  // Tile counters of the step's four persistent MLP launches (forward / backward(data) of each level), zeroed by the step's
  // first launch: the dense kernels take their tiles from them when PXO_TUNE_TILE_SCHED = 1, the skipping backward always
  unsigned int* const counters = reinterpret_cast<unsigned int*>(t.scalars + 100);
  const bool dyn = tune_tile_sched() != 0;
  unsigned int* const cnt_fwd_c = dyn ? counters + 0 : nullptr;
  unsigned int* const cnt_fwd_f = dyn ? counters + 2 : nullptr;
  unsigned int* const cnt_bwd_c = (dyn || cfg->skip_zero_rows) ? counters + 1 : nullptr;
  unsigned int* const cnt_bwd_f = (dyn || cfg->skip_zero_rows) ? counters + 3 : nullptr;
  const int Nf = cfg->num_fine_samples;
  Draws dr;
  PXO_TRY(prepare_draws(cfg, t, B, randomized, t_rand, u, sp_points, seed, s, dr, params, 2 * n_mlp, sumsq_partial, counters, 4));
  // coarse level: forward, losses (train.py:77-98), reverse of the compositing, reverse through MLP_0.  Nothing of the fine
  // level feeds MLP_0's gradient (the fine sample positions carry no gradient, model_utils.py:286), so it is complete here
  // -- a quarter into the step -- and its all-reduce can ride under the fine level.
  PXO_TRY(forward_coarse(cfg, t, packed_fwd0, origins, directions, viewdirs, B, dr, pixels, nullptr, nullptr, nullptr, s,
                         cnt_fwd_c));
  // skip_zero_rows: rows with an exactly zero upstream gradient are left out of the reverse pass (bit-identical gradients)
  // (a pass whose weight-gradient row ranges would not fit the kernels' live-chunk lists -- > 32,768 rows per range, i.e.
  // more than 8.4 M sample rows on 256 CUs -- makes the whole reverse pass dense: same bits, no saving)
  const bool skip = cfg->skip_zero_rows != 0 && wgrad_skip_supported(t.c.M) && (Nf == 0 || wgrad_skip_supported(t.f.M));
  {
    std::lock_guard<std::mutex> lk(g_step_mu);
    g_step_skipped[ws] = skip;
  }
  uint8_t* const live_c = skip ? t.c.live : nullptr;
  uint8_t* const live_f = skip ? t.f.live : nullptr;
  // The coarse reverse pass touches nothing the fine forward reads or writes (its own workspace slices, the MLP_0 half of
  // `grads`; the weight-gradient slabs are shared with the fine pass, which therefore waits for it): with a fine level it runs
  // on the side stream, beside sample_pdf / the fine forward, and joins before the fine weight gradients.
  // bf16x6 with its own weight-gradient kernel: that kernel also sums the columns of dz_1..7 (the bias gradients of Dense_1..7)
  // while it streams them, and backward(data) leaves its per-layer lane reductions for those layers out (read ONCE per step: the
  // two launches of a pass must agree)
  const int bias_flags = cfg->mlp_precision == PXO_MLP_BF16X6 && tune_x6_wgrad() != 0 ? kBiasFromWgrad : 0;
  const bool fork = Nf > 0 && g_tune_coarse_stream.load() != 0 && side_stream_ready();
  hipStream_t sc = s;
  if (fork) {
    if (hipEventRecord(g_ev_fork, s) != hipSuccess || hipStreamWaitEvent(g_side_stream, g_ev_fork, 0) != hipSuccess) {
      set_error("pxo_train_fwd_bwd: fork to the side stream failed");
      return PXO_ERR_HIP;
    }
    sc = g_side_stream;
  }
  PXO_TRY(launch_mlp_bwd_data(cfg, packed_bwd0, t.c.d_raw_rgb, t.c.d_raw_sigma, t.c.mask, t.c.M, t.c.dz, t.c.dbias, live_c,
                              cnt_bwd_c, sc, true, bias_flags));
  PXO_TRY(launch_mlp_bwd_weights(cfg, t.c.acts, t.c.enc, t.c.dz, t.c.d_raw_rgb, t.c.d_raw_sigma, t.c.dbias, t.c.M,
                                 grads, t.wgrad_ws, t.wgrad_bytes, live_c, sc, bias_flags));
  if (cfg->weight_decay_mult != 0.f) PXO_TRY(launch_axpy(grads, params, n_mlp, wd_coef, sc));
  if (grads0_ready && hipEventRecord((hipEvent_t)grads0_ready, sc) != hipSuccess) {
    set_error("pxo_train_fwd_bwd: hipEventRecord(grads0_ready) failed");
    return PXO_ERR_HIP;
  }
  if (fork && hipEventRecord(g_ev_join, sc) != hipSuccess) {
    set_error("pxo_train_fwd_bwd: hipEventRecord(join) failed");
    return PXO_ERR_HIP;
  }
  if (Nf > 0) {
    PXO_TRY(forward_fine(cfg, t, packed_fwd1, origins, directions, viewdirs, B, dr, pixels, nullptr, nullptr, nullptr, s,
                         cnt_fwd_f));
    PXO_TRY(launch_mlp_bwd_data(cfg, packed_bwd1, t.f.d_raw_rgb, t.f.d_raw_sigma, t.f.mask, t.f.M, t.f.dz, t.f.dbias, live_f,
                                cnt_bwd_f, s, true, bias_flags));
    if (fork && hipStreamWaitEvent(s, g_ev_join, 0) != hipSuccess) {       // the slabs are the coarse pass's until here
      set_error("pxo_train_fwd_bwd: join of the side stream failed");
      return PXO_ERR_HIP;
    }
    PXO_TRY(launch_mlp_bwd_weights(cfg, t.f.acts, t.f.enc, t.f.dz, t.f.d_raw_rgb, t.f.d_raw_sigma, t.f.dbias, t.f.M,
                                   grads + n_mlp, t.wgrad_ws, t.wgrad_bytes, live_f, s, bias_flags));
  } else {
    PXO_TRY(launch_fill(grads + n_mlp, n_mlp, 0.f, s));
  }
  if (cfg->weight_decay_mult != 0.f) PXO_TRY(launch_axpy(grads + n_mlp, params + n_mlp, n_mlp, wd_coef, s));
  PXO_TRY(launch_finalize_stats(Nf > 0 ? t.f.ray_sse : nullptr, t.c.ray_sse, t.n_sp > 0 ? t.sp_exp : nullptr, sumsq_partial,
                                B, t.n_sp, cfg->sparsity_weight, 2 * n_mlp, stats, s));
  return PXO_OK;
}

int pxo_train_fwd_bwd(const PxoCfg* cfg, const float* params, const float* packed_fwd0, const float* packed_bwd0,
                      const float* packed_fwd1, const float* packed_bwd1, const float* origins,
                      const float* directions, const float* viewdirs, const float* pixels, int64_t B, int randomized,
                      const float* t_rand, const float* u, const float* sp_points, uint64_t seed, float* grads,
                      float* stats, void* ws, size_t ws_bytes, void* stream) {
  return pxo_train_fwd_bwd_bucketed(cfg, params, packed_fwd0, packed_bwd0, packed_fwd1, packed_bwd1, origins, directions,
                                    viewdirs, pixels, B, randomized, t_rand, u, sp_points, seed, grads, stats, ws, ws_bytes,
                                    nullptr, stream);
}

int pxo_train_backward_work(const PxoCfg* cfg, int64_t B, void* ws, size_t ws_bytes, int64_t* live_chunks,
                            int64_t* total_chunks, void* stream) {
  PXO_TRY(validate_cfg(cfg));
  PXO_REQUIRE(B >= 1 && ws && live_chunks && total_chunks, "pxo_train_backward_work: bad arguments");
  TrainWs t;
  carve_train(cfg, B, ws, true, t);
  if (ws_bytes < t.total) { set_error("pxo_train_backward_work: workspace %zu < %zu", ws_bytes, t.total); return PXO_ERR_WORKSPACE; }
  hipStream_t s = (hipStream_t)stream;
  const int64_t nc = (t.c.M + kLiveRows - 1) / kLiveRows, nf = cfg->num_fine_samples > 0 ? (t.f.M + kLiveRows - 1) / kLiveRows : 0;
  *total_chunks = nc + nf;
  // dense pass (also when the step fell back to it because a row range would not fit the live lists): every chunk is live.
  // The mode is the one the last step on this workspace RECORDED, not one re-derived from today's tuning knobs.
  bool skipped = false;
  {
    std::lock_guard<std::mutex> lk(g_step_mu);
    auto it = g_step_skipped.find(ws);
    if (it == g_step_skipped.end()) { set_error("pxo_train_backward_work: no pxo_train_fwd_bwd call has used this workspace"); return PXO_ERR_ARG; }
    skipped = it->second;
  }
  if (!skipped) {
    *live_chunks = nc + nf;
    return PXO_OK;
  }
  unsigned long long* cnt = reinterpret_cast<unsigned long long*>(t.scalars + 96);     // 8-byte aligned tail of the scalars block
  if (hipMemsetAsync(cnt, 0, sizeof(unsigned long long), s) != hipSuccess) { set_error("hipMemsetAsync failed"); return PXO_ERR_HIP; }
  PXO_TRY(launch_count_live(t.c.live, nc, cnt, s));
  if (nf > 0) PXO_TRY(launch_count_live(t.f.live, nf, cnt, s));
  unsigned long long host = 0;
  if (hipMemcpyAsync(&host, cnt, sizeof(host), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
    set_error("pxo_train_backward_work: copy back failed");
    return PXO_ERR_HIP;
  }
  *live_chunks = (int64_t)host;
  return PXO_OK;
}

// plain handles for the bucketed form: the caller's runtime (torch) creates its events lazily and keeps them private
int pxo_event_create(void** event) {
  PXO_REQUIRE(event != nullptr, "pxo_event_create: NULL pointer");
  hipEvent_t e = nullptr;
  if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { set_error("hipEventCreateWithFlags failed"); return PXO_ERR_HIP; }
  *event = (void*)e;
  return PXO_OK;
}
int pxo_event_destroy(void* event) {
  if (event && hipEventDestroy((hipEvent_t)event) != hipSuccess) { set_error("hipEventDestroy failed"); return PXO_ERR_HIP; }
  return PXO_OK;
}
int pxo_stream_wait_event(void* stream, void* event) {
  PXO_REQUIRE(event != nullptr, "pxo_stream_wait_event: NULL event");
  if (hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0) != hipSuccess) { set_error("hipStreamWaitEvent failed"); return PXO_ERR_HIP; }
  return PXO_OK;
}

int pxo_adam_pack_step(const PxoCfg* cfg, float* params, float* m, float* v, const float* grads, float lr, int64_t step,
                       float grad_scale, float* packed_fwd0, float* packed_bwd0, float* packed_fwd1, float* packed_bwd1,
                       void* stream) {
  PXO_TRY(validate_cfg(cfg));
  PXO_REQUIRE(params && m && v && grads && step >= 0 && packed_fwd0 && packed_fwd1, "pxo_adam_pack_step: bad arguments");
  PXO_REQUIRE((packed_bwd0 != nullptr) == (packed_bwd1 != nullptr), "pxo_adam_pack_step: both backward images or none");
  if (cfg->mlp_precision != PXO_MLP_F32) {
    // split-precision images: the Adam kernel, then that precision's packing kernels (same results as the separate calls)
    if (cfg->mlp_precision == PXO_MLP_BF16X3 && packed_bwd0) {
      set_error("pxo_adam_pack_step: the bf16x3 images are forward-only (packed_bwd must be NULL)");
      return PXO_ERR_UNSUPPORTED;
    }
    const int64_t n_mlp = mlp_param_count(cfg->sh_deg);
    PXO_TRY(launch_adam(params, m, v, grads, 2 * n_mlp, lr, step, grad_scale, (hipStream_t)stream));
    PXO_TRY(launch_pack(cfg, params, packed_fwd0, packed_bwd0, (hipStream_t)stream));
    return launch_pack(cfg, params + n_mlp, packed_fwd1, packed_bwd1, (hipStream_t)stream);
  }
  return launch_adam_pack(cfg, params, m, v, grads, lr, step, grad_scale, packed_fwd0, packed_bwd0, packed_fwd1,
                          packed_bwd1, (hipStream_t)stream);
}

int pxo_occupy_cus(int blocks, int threads, float micros, int lds_bytes, void* stream) {
  PXO_REQUIRE(blocks >= 1 && blocks <= 4096 && threads >= 64 && threads <= 1024 && threads % 64 == 0 && micros >= 0.f &&
                  micros <= 1e6f && lds_bytes >= 0 && lds_bytes <= 65536,
              "pxo_occupy_cus: bad arguments (blocks %d, threads %d, micros %g, lds_bytes %d)", blocks, threads, (double)micros,
              lds_bytes);
  int dev = 0, khz = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0)
    khz = 100000;                                                    // gfx9: 100 MHz constant clock
  const long long ticks = (long long)((double)micros * 1e-3 * (double)khz);
  hipLaunchKernelGGL(occupy_kernel, dim3(blocks), dim3(threads), (size_t)lds_bytes, (hipStream_t)stream, ticks);
  return check_launch("occupy_cus");
}

int pxo_profile_enable(int tag_mask) {
  PXO_REQUIRE(tag_mask >= 0 && tag_mask < (1 << PXO_PROF_NUM_TAGS), "pxo_profile_enable: mask %d has bits beyond the %d tags",
              tag_mask, PXO_PROF_NUM_TAGS);
  g_prof_mask = (unsigned)tag_mask;
  return PXO_OK;
}

int pxo_profile_read(int tag, int64_t* launches, double* total_ms, int64_t* total_rows) {
  PXO_REQUIRE(tag >= 0 && tag < PXO_PROF_NUM_TAGS && launches && total_ms && total_rows, "pxo_profile_read: bad arguments");
  int64_t n = 0, rows = 0;
  double ms = 0.0;
  std::vector<ProfRecord> keep;
  for (auto& r : g_prof) {
    if (r.tag != tag) { keep.push_back(r); continue; }
    float t = 0.f;
    if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) {
      ++n; ms += t; rows += r.rows;
    }
    (void)hipEventDestroy(r.a);
    (void)hipEventDestroy(r.b);
  }
  g_prof.swap(keep);
  *launches = n; *total_ms = ms; *total_rows = rows;
  return PXO_OK;
}

int pxo_eval_points(const PxoCfg* cfg, const float* packed_fwd, const float* points, int64_t N, float* raw_rgb,
                    float* raw_sigma, void* stream) {
  PXO_TRY(validate_cfg(cfg));
  if (N == 0) return PXO_OK;
  PXO_REQUIRE(N >= 0 && packed_fwd && points && raw_sigma, "pxo_eval_points: bad arguments");
  return launch_mlp_fwd(cfg, packed_fwd, points, N, raw_rgb, raw_sigma, nullptr, nullptr, nullptr,
                        (hipStream_t)stream);
}

int pxo_grid_sigma(const PxoCfg* cfg, const float* packed_fwd, int reso, int x0, int x1, const float offset[3],
                   const float scale[3], float* sigma_out, void* stream) {
  PXO_TRY(validate_cfg(cfg));
  PXO_REQUIRE(reso >= 1 && x0 >= 0 && x1 >= x0 && x1 <= reso, "pxo_grid_sigma: bad slab [%d,%d) of %d", x0, x1, reso);
  if (x1 == x0) return PXO_OK;
  PXO_REQUIRE(packed_fwd && offset && scale && sigma_out, "pxo_grid_sigma: NULL pointer");
  return launch_mlp_fwd_grid(cfg, packed_fwd, reso, x0, x1, offset, scale, sigma_out, (hipStream_t)stream);
}

}  // extern "C"
