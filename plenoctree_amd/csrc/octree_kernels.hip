// PlenOctree side of the path on gfx950: N3Tree build from the grid mask, leaf sampling, the
// training-view weight mask over the dense grid, and the octree volume renderer (forward + gradient).
// C ABI: include/plenoctree_octree.h (reference call sites cited there).
//
// This translation unit is compiled with -ffp-contract=off: the marching arithmetic (sample positions,
// cell exits, step lengths) is written one rounding per operation, so that the sequence of cells a ray
// visits does not depend on how the compiler fuses multiply-adds (oracle/octree_oracle.py takes the
// same steps in numpy float32).
//
// Tree build: with whole levels refined at a time (octree/extraction.py:341-350), svox's node order --
// breadth-first, each level sorted by packed parent cell index -- is the Morton (z-curve) order of the
// occupied cells of that level.  So the build is: occupancy pyramid stored in Morton order, one exclusive
// scan per level (rank of a cell = its node index inside the level), and the 8 children of cell m are the
// contiguous cells 8m..8m+7 of the next level.  No sort, no pointer chasing, all streams are linear.
//
// Renderer: one ray per row of ROW = 8 or 16 lanes of the wave (8 or 4 rays per wave64).  The lanes of a row walk
// the tree together (uniform control flow inside a row), each lane owns data channels lane, lane+ROW, ... of the
// leaf, so a leaf's 3*K coefficients are fetched as coalesced 32/64-byte row loads and reduced with row-local
// DPP shuffles.  The node path of the previous sample is kept in LDS; the descent for the next sample
// resumes at the deepest node the two positions share instead of the root.
#include "pxo_common.h"
#include "pxo_sh.h"
#include "../../include/plenoctree_octree.h"

namespace pxo {

constexpr int kMaxD = PXO_TREE_MAX_DEPTH;
constexpr int kBits = kMaxD + 1;             // bits per axis of the finest cell grid (2^(depth+1))

// ------------------------------------------------------------------------------------------
// Morton helpers: cell (x,y,z) of a 2^d grid <-> m, x the most significant bit of each triple
// (cell index inside a node = (i*2+j)*2+k, svox packed order).
// ------------------------------------------------------------------------------------------
__host__ __device__ inline void morton_decode(uint64_t m, int d, uint32_t& x, uint32_t& y, uint32_t& z) {
  x = y = z = 0;
  for (int b = 0; b < d; ++b) {
    const uint32_t t = (uint32_t)(m >> (3 * b)) & 7u;
    x |= ((t >> 2) & 1u) << b;
    y |= ((t >> 1) & 1u) << b;
    z |= (t & 1u) << b;
  }
}

__host__ inline int64_t pow8(int d) { return (int64_t)1 << (3 * d); }

// workspace carving (bytes): occ[1..depth], rank[1..depth], block sums, counts
struct TreeWs {
  int64_t occ_off[kMaxD + 1];
  int64_t rank_off[kMaxD + 1];
  int64_t bsum_off, count_off, total;
};
constexpr int kScanElems = 2048;             // elements per scan block (256 threads x 8 bytes)

static TreeWs tree_ws(int depth) {
  TreeWs w{};
  int64_t off = 0;
  auto take = [&](int64_t bytes) { int64_t o = off; off += (bytes + 255) & ~(int64_t)255; return o; };
  for (int d = 1; d <= depth; ++d) w.occ_off[d] = take(pow8(d));
  for (int d = 1; d <= depth; ++d) w.rank_off[d] = take(4 * pow8(d));
  w.bsum_off = take(4 * (pow8(depth) / kScanElems + 2));
  w.count_off = take(8 * (kMaxD + 2));
  w.total = off;
  return w;
}

__global__ void threshold_mask_kernel(const float* __restrict__ v, int64_t n, float thresh, uint8_t* __restrict__ mask) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) mask[i] = v[i] >= thresh ? 1 : 0;
}

// level `depth` occupancy (Morton order) from the 2^(depth+1) mask (x slowest)
__global__ void pyramid_base_kernel(const uint8_t* __restrict__ mask, int depth, uint8_t* __restrict__ occ) {
  const int64_t m = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (m >= ((int64_t)1 << (3 * depth))) return;
  uint32_t x, y, z;
  morton_decode((uint64_t)m, depth, x, y, z);
  const int64_t reso = (int64_t)2 << depth;
  uint32_t any = 0;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int64_t base = ((int64_t)(2 * x + (c >> 1)) * reso + (2 * y + (c & 1))) * reso + 2 * z;
    any |= *reinterpret_cast<const uint16_t*>(mask + base);   // 2z is even, reso even: 2-byte aligned
  }
  occ[m] = any ? 1 : 0;
}

__global__ void pyramid_up_kernel(const uint8_t* __restrict__ fine, int64_t n_coarse, uint8_t* __restrict__ coarse) {
  const int64_t m = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (m >= n_coarse) return;
  coarse[m] = reinterpret_cast<const uint64_t*>(fine)[m] ? 1 : 0;
}

__device__ __forceinline__ uint32_t bytes_sum(uint64_t v) {       // bytes are 0/1
  return (uint32_t)__popcll(v);
}

// block-level exclusive scan of 256 per-thread counts
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* s_wave, uint32_t& block_total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t up = __shfl_up(inc, o);
    if (lane >= o) inc += up;
  }
  if (lane == 63) s_wave[wave] = inc;
  __syncthreads();
  uint32_t base = 0, tot = 0;
  for (int w = 0; w < 4; ++w) {
    if (w < wave) base += s_wave[w];
    tot += s_wave[w];
  }
  block_total = tot;
  return base + inc - v;
}

__global__ __launch_bounds__(256) void scan_blocksum_kernel(const uint8_t* __restrict__ occ, int64_t n,
                                                             uint32_t* __restrict__ bsum) {
  __shared__ uint32_t s_wave[4];
  const int64_t i8 = (blockIdx.x * (int64_t)256 + threadIdx.x) * 8;
  const uint32_t v = i8 < n ? bytes_sum(*reinterpret_cast<const uint64_t*>(occ + i8)) : 0;
  uint32_t tot;
  block_excl_scan(v, s_wave, tot);
  if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}

// single block: in-place exclusive scan of nb block sums; total -> *count
__global__ __launch_bounds__(256) void scan_sums_kernel(uint32_t* __restrict__ bsum, int64_t nb, int64_t* __restrict__ count) {
  __shared__ uint32_t s_wave[4];
  uint32_t carry = 0;
  for (int64_t base = 0; base < nb; base += 256) {
    const int64_t i = base + threadIdx.x;
    const uint32_t v = i < nb ? bsum[i] : 0;
    uint32_t tot;
    const uint32_t ex = block_excl_scan(v, s_wave, tot);
    if (i < nb) bsum[i] = carry + ex;
    carry += tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) *count = carry;
}

__global__ __launch_bounds__(256) void scan_final_kernel(const uint8_t* __restrict__ occ, int64_t n,
                                                          const uint32_t* __restrict__ bsum, int32_t* __restrict__ rank) {
  __shared__ uint32_t s_wave[4];
  const int64_t i8 = (blockIdx.x * (int64_t)256 + threadIdx.x) * 8;
  const uint64_t bytes = i8 < n ? *reinterpret_cast<const uint64_t*>(occ + i8) : 0;
  uint32_t tot;
  uint32_t run = bsum[blockIdx.x] + block_excl_scan(bytes_sum(bytes), s_wave, tot);
  if (i8 >= n) return;
  int32_t r[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    r[k] = (int32_t)run;
    run += (uint32_t)(bytes >> (8 * k)) & 1u;
  }
  reinterpret_cast<int4*>(rank + i8)[0] = make_int4(r[0], r[1], r[2], r[3]);
  reinterpret_cast<int4*>(rank + i8)[1] = make_int4(r[4], r[5], r[6], r[7]);
}

// nodes of depth d (cells m of level d that are occupied; d = 0: the root)
struct LevelPtrs {
  const uint8_t* occ[kMaxD + 2];
  const int32_t* rank[kMaxD + 2];
  int64_t start[kMaxD + 2];
};

__global__ void tree_emit_kernel(LevelPtrs lv, int d, int depth, int32_t* __restrict__ child,
                                 int32_t* __restrict__ parent_depth) {
  const int64_t m = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (m >= ((int64_t)1 << (3 * d))) return;
  if (d > 0 && !lv.occ[d][m]) return;
  const int64_t n = d == 0 ? 0 : lv.start[d] + lv.rank[d][m];
  int32_t ch[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    ch[c] = 0;
    if (d < depth) {
      const int64_t mc = m * 8 + c;
      if (lv.occ[d + 1][mc]) ch[c] = (int32_t)(lv.start[d + 1] + lv.rank[d + 1][mc] - n);
    }
  }
  reinterpret_cast<int4*>(child + n * 8)[0] = make_int4(ch[0], ch[1], ch[2], ch[3]);
  reinterpret_cast<int4*>(child + n * 8)[1] = make_int4(ch[4], ch[5], ch[6], ch[7]);
  int64_t packed = 0;
  if (d > 0) {
    const int64_t pm = m >> 3;
    const int64_t pn = d == 1 ? 0 : lv.start[d - 1] + lv.rank[d - 1][pm];
    packed = pn * 8 + (m & 7);
  }
  parent_depth[n * 2] = (int32_t)packed;
  parent_depth[n * 2 + 1] = d;
}

// ------------------------------------------------------------------------------------------
// leaf samples
// ------------------------------------------------------------------------------------------
__global__ void tree_sample_cells_kernel(const int32_t* __restrict__ parent_depth, int64_t node0, int64_t n_nodes,
                                         int S, const float* __restrict__ u, float ox, float oy, float oz,
                                         float ix, float iy, float iz, float* __restrict__ pts) {
  const int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;      // (node, cell, sample)
  if (q >= n_nodes * 8 * S) return;
  const int64_t cellq = q / S;
  int64_t node = node0 + (cellq >> 3);
  const int cell = (int)(cellq & 7);
  const int depth = parent_depth[node * 2 + 1];
  uint32_t X = (cell >> 2) & 1, Y = (cell >> 1) & 1, Z = cell & 1;
  for (int lvl = 1; lvl <= depth; ++lvl) {
    const int32_t packed = parent_depth[node * 2];
    const int c = packed & 7;
    X |= (uint32_t)((c >> 2) & 1) << lvl;
    Y |= (uint32_t)((c >> 1) & 1) << lvl;
    Z |= (uint32_t)(c & 1) << lvl;
    node = packed >> 3;
  }
  const float side = 1.0f / (float)((uint32_t)2 << depth);
  const float px = (float)X * side + u[q * 3] * side;
  const float py = (float)Y * side + u[q * 3 + 1] * side;
  const float pz = (float)Z * side + u[q * 3 + 2] * side;
  pts[q * 3] = (px - ox) / ix;
  pts[q * 3 + 1] = (py - oy) / iy;
  pts[q * 3 + 2] = (pz - oz) / iz;
}

// tree[leaf_inds].sample(S) for arbitrary leaves given by their packed cell index (node*8 + cell)
__global__ void tree_sample_leaves_kernel(const int32_t* __restrict__ parent_depth, const int64_t* __restrict__ packed,
                                          int64_t n_cells, int S, const float* __restrict__ u, float ox, float oy, float oz,
                                          float ix, float iy, float iz, float* __restrict__ pts) {
  const int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;      // (cell, sample)
  if (q >= n_cells * S) return;
  const int64_t pk = packed[q / S];
  int64_t node = pk >> 3;
  const int cell = (int)(pk & 7);
  const int depth = parent_depth[node * 2 + 1];
  uint32_t X = (cell >> 2) & 1, Y = (cell >> 1) & 1, Z = cell & 1;
  for (int lvl = 1; lvl <= depth; ++lvl) {
    const int32_t up = parent_depth[node * 2];
    const int c = up & 7;
    X |= (uint32_t)((c >> 2) & 1) << lvl;
    Y |= (uint32_t)((c >> 1) & 1) << lvl;
    Z |= (uint32_t)(c & 1) << lvl;
    node = up >> 3;
  }
  const float side = 1.0f / (float)((uint32_t)2 << depth);
  const float px = (float)X * side + u[q * 3] * side;
  const float py = (float)Y * side + u[q * 3 + 1] * side;
  const float pz = (float)Z * side + u[q * 3 + 2] * side;
  pts[q * 3] = (px - ox) / ix;
  pts[q * 3 + 1] = (py - oy) / iy;
  pts[q * 3 + 2] = (pz - oz) / iz;
}

// tree[points]: packed index (node*8 + cell) of the leaf containing each world-space point.  svox's query: tree
// coordinates clamped to [0, 1 - 1e-6], then per level x *= 2, cell = floor(x), x -= cell until a leaf is reached.
__global__ void tree_query_kernel(const int32_t* __restrict__ child, const float* __restrict__ pts, int64_t n, float ox,
                                  float oy, float oz, float ix, float iy, float iz, int64_t* __restrict__ packed) {
  const int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (q >= n) return;
  float x = fminf(fmaxf(ox + ix * pts[q * 3], 0.0f), 1.0f - 1e-6f);
  float y = fminf(fmaxf(oy + iy * pts[q * 3 + 1], 0.0f), 1.0f - 1e-6f);
  float z = fminf(fmaxf(oz + iz * pts[q * 3 + 2], 0.0f), 1.0f - 1e-6f);
  int64_t node = 0;
  for (int lvl = 0; lvl <= kMaxD + 1; ++lvl) {
    x *= 2.0f; y *= 2.0f; z *= 2.0f;
    const int i = (int)floorf(x), j = (int)floorf(y), k = (int)floorf(z);
    x -= (float)i; y -= (float)j; z -= (float)k;
    const int cell = (i * 2 + j) * 2 + k;
    const int32_t skip = child[node * 8 + cell];
    if (skip == 0) {
      packed[q] = node * 8 + cell;
      return;
    }
    node += skip;
  }
  packed[q] = -1;    // deeper than any valid tree: corrupt child array
}

__global__ void tree_relu_sigma_kernel(float* __restrict__ data, int64_t n_cells, int dim) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n_cells) {
    float& s = data[i * dim + dim - 1];
    s = fmaxf(s, 0.0f);
  }
}

// ------------------------------------------------------------------------------------------
// ray set-up shared by the marchers (svox cam2world_ray / transform_coord / _get_delta_scale / _dda_unit)
// ------------------------------------------------------------------------------------------
struct TreeRay {
  float o[3], d[3], invdir[3], vdir[3];
  float delta_scale, tmin, tmax;
};

__device__ __forceinline__ void dda_unit(const float (&cen)[3], const float (&invdir)[3], float& tmin, float& tmax) {
  tmin = 0.0f;
  tmax = 1e9f;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float t1 = -cen[a] * invdir[a];
    const float t2 = t1 + invdir[a];
    tmin = fmaxf(tmin, fminf(t1, t2));
    tmax = fminf(tmax, fmaxf(t1, t2));
  }
}

__device__ __forceinline__ void camera_ray(const float* __restrict__ c2w, float fx, float fy, int W, int H, int px, int py,
                                           float (&origin)[3], float (&dir)[3]) {
  float x = ((float)px - 0.5f * (float)W) / fx;
  float y = -((float)py - 0.5f * (float)H) / fy;
  float z = sqrtf((x * x + y * y) + 1.0f);
  x = x / z;
  y = y / z;
  z = -1.0f / z;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    dir[a] = (c2w[a * 4] * x + c2w[a * 4 + 1] * y) + c2w[a * 4 + 2] * z;
    origin[a] = c2w[a * 4 + 3];
  }
}

__device__ __forceinline__ void to_tree_ray(const float (&origin)[3], const float (&dir)[3], const float* offset,
                                            const float* invradius, TreeRay& r) {
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    r.o[a] = offset[a] + invradius[a] * origin[a];
    r.d[a] = dir[a] * invradius[a];
  }
  const float nrm = sqrtf((r.d[0] * r.d[0] + r.d[1] * r.d[1]) + r.d[2] * r.d[2]);
  r.delta_scale = 1.0f / nrm;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    r.d[a] = r.d[a] * r.delta_scale;
    r.invdir[a] = 1.0f / (r.d[a] + 1e-9f);
  }
  dda_unit(r.o, r.invdir, r.tmin, r.tmax);
}

__device__ __forceinline__ float clamp_coord(float x) { return fminf(fmaxf(x, 0.0f), 1.0f - 1e-6f); }

// dda_unit for a point INSIDE the unit cell (0 <= cen < 1), as every march step calls it.  The entry distance is then
// max(0, non-positive values) = 0, and per axis the exit is max(t1, t2) with t1 = -cen * invdir, t2 = t1 + invdir: t2 when
// invdir >= 0, t1 otherwise - i.e. t1 + add with the per-ray addend add = invdir >= 0 ? invdir : 0 (t1 + 0 is t1).  Same
// operations, same roundings as dda_unit (this file is compiled without contraction), 9 instructions instead of 19.
// Returns tmax (= tmax - tmin).
// init(invdir, s) with s a POWER OF TWO returns s x the exit distance instead: scaling by 2^k commutes with every rounding of
// the mul / add / min chain (no operand gets near the under- or overflow range), so `cell_exit(local) * 2^-k` of the
// power-of-two weight-mask marchers is folded into the per-ray constants -- bit for bit the same step length, one
// multiplication per sample less.
struct CellExit {
  float inv[3], add[3], cap;
  __device__ __forceinline__ void init(const float (&invdir)[3], float pow2_scale = 1.0f) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      inv[a] = invdir[a] * pow2_scale;
      add[a] = invdir[a] >= 0.0f ? invdir[a] * pow2_scale : 0.0f;
    }
    cap = 1e9f * pow2_scale;
  }
  __device__ __forceinline__ float operator()(const float (&cen)[3]) const {
    const float e0 = -cen[0] * inv[0] + add[0];
    const float e1 = -cen[1] * inv[1] + add[1];
    const float e2 = -cen[2] * inv[2] + add[2];
    return fminf(fminf(cap, e0), fminf(e1, e2));
  }
};
// 2^-(depth+1) exactly: dividing the cell-local distance by the cell count per axis (a power of two) is this multiplication
__device__ __forceinline__ float inv_cells(int depth) { return __int_as_float((126 - depth) << 23); }

// ------------------------------------------------------------------------------------------
// grid weight render: one thread per (camera, pixel)
// ------------------------------------------------------------------------------------------
struct Vec3 { float v[3]; };

// sigma and the weights are walked in 4x4x4 bricks (256 B = two cache lines per brick): a marching ray takes
// several samples per brick and the 8x8-pixel footprint of a wave spans only a few bricks, whereas in the
// x-slowest layout nearly every sample of every lane touches its own line (PMC: 72 GB fetched per two
// cameras against 4.6 GB of 4-byte samples; profiles/r01_octree_kernels.md).
__host__ __device__ inline int64_t brick_index(int x, int y, int z, int nb) {
  return ((((int64_t)(x >> 2) * nb + (y >> 2)) * nb + (z >> 2)) << 6) | ((x & 3) << 4) | ((y & 3) << 2) | (z & 3);
}

// also counts the voxels above sigma_thresh: the weight-mask launch picks its kernel by that fraction.  Grid-stride loop, one
// atomic per wave at the END (an atomic per wave and brick -- 2 M of them on one address at 512^3 -- took 25 ms)
__global__ __launch_bounds__(256) void brick_sigma_kernel(const float* __restrict__ lin, int reso, float* __restrict__ bricked,
                                                           float sigma_thresh, unsigned long long* __restrict__ occupied) {
  const int64_t n = (int64_t)reso * reso * reso;
  const int nb = reso >> 2;
  unsigned long long count = 0;                // wave-uniform
  for (int64_t i0 = blockIdx.x * (int64_t)256; i0 < n; i0 += (int64_t)gridDim.x * 256) {   // n is a multiple of 64: waves stay whole
    const int64_t i = i0 + threadIdx.x;       // bricked index
    float v = -INFINITY;
    if (i < n) {
      const int l = (int)(i & 63);
      const int64_t b = i >> 6;
      const int bz = (int)(b % nb), by = (int)((b / nb) % nb), bx = (int)(b / ((int64_t)nb * nb));
      const int x = bx * 4 + (l >> 4), y = by * 4 + ((l >> 2) & 3), z = bz * 4 + (l & 3);
      v = lin[((int64_t)x * reso + y) * reso + z];
      bricked[i] = v;
    }
    count += (unsigned long long)__builtin_popcountll(__builtin_amdgcn_ballot_w64(v > sigma_thresh));
  }
  if ((threadIdx.x & 63) == 0 && count) atomicAdd(occupied, count);
}

__global__ void unbrick_max_kernel(const float* __restrict__ bricked, int reso, float* __restrict__ lin) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;       // linear index
  if (i >= (int64_t)reso * reso * reso) return;
  const int z = (int)(i % reso), y = (int)((i / reso) % reso), x = (int)(i / ((int64_t)reso * reso));
  lin[i] = fmaxf(lin[i], bricked[brick_index(x, y, z, reso >> 2)]);
}

// Which of the two weight-mask marchers runs is decided ON THE DEVICE (the call stays asynchronous; one launch, one uniform
// branch at its top).  Measured per 800x800 camera at 512^3, per-sample / slab-staged: density with fuzzy tails (97 % of the
// voxels above the threshold) 1.97 / 1.29 ms; the same spheres with exact zeros outside (6 % above) 0.95 / 1.08 ms; a NeRF
// after 105 steps (18 % above) 0.89 / 0.97 ms -- the slab marcher's staging, barriers and flush are paid per brick layer
// whether or not the rays find anything in it.
// Which 16x16-pixel tile of which camera a workgroup of the weight-mask launch takes.
//   order 0  linear: workgroup b -> camera b / tiles, tile b % tiles (row-major); consecutive workgroups go round-robin over
//            the 8 XCDs, so neighbouring tiles -- which cross the same bricks -- sit behind eight different L2s (PMC: 2.2 GB
//            read per camera against a floor of 0.46 GB)
//   order 1  XCD supertiles: the tiles of a 4x4-tile square (64x64 pixels) all go to ONE XCD (workgroup ids that are equal
//            mod 8), the squares round-robin over the XCDs, so an L2 sees whole squares and every XCD still gets every
//            part of the image.  The grid is padded (8 * ceil(squares / 8) * 16 workgroups per camera); padding returns at once.
struct GwTile { int cam, tile; };
__device__ __forceinline__ GwTile gw_tile(int64_t b, int tiles_x, int tiles_y, int order) {
  if (order == 0) {
    const int64_t tiles = (int64_t)tiles_x * tiles_y;
    return GwTile{(int)(b / tiles), (int)(b % tiles)};
  }
  const int super_x = (tiles_x + 3) >> 2, super_y = (tiles_y + 3) >> 2, nsuper = super_x * super_y;
  const int64_t per_cam = (int64_t)8 * ((nsuper + 7) >> 3) * 16;
  const int cam = (int)(b / per_cam), r = (int)(b % per_cam);
  const int xcd = r & 7, idx = r >> 3, sq = xcd + 8 * (idx >> 4), within = idx & 15;
  if (sq >= nsuper) return GwTile{cam, -1};
  const int tx = (sq % super_x) * 4 + (within & 3), ty = (sq / super_x) * 4 + (within >> 2);
  if (tx >= tiles_x || ty >= tiles_y) return GwTile{cam, -1};
  return GwTile{cam, ty * tiles_x + tx};
}
__host__ inline int64_t gw_blocks_per_cam(int tiles_x, int tiles_y, int order) {
  if (order == 0) return (int64_t)tiles_x * tiles_y;
  const int nsuper = ((tiles_x + 3) >> 2) * ((tiles_y + 3) >> 2);
  return (int64_t)8 * ((nsuper + 7) >> 3) * 16;
}

struct GwSelect {
  const unsigned long long* occupied;   // voxels above sigma_thresh (brick_sigma_kernel)
  int64_t n;                            // voxels
  int force;                            // 1: slab-staged, 0: per-sample, -1: by the occupied fraction
  int order;                            // gw_tile: 0 linear, 1 XCD supertiles
  __device__ __forceinline__ bool dense() const {
    return force >= 0 ? force != 0 : 2 * (int64_t)*occupied > n;      // more than half of the grid above the threshold
  }
};

// POW2: reso is a power of two (every grid the extraction makes): the division by it is an exact multiplication and
// the brick index is assembled from bit fields in 32 bits (reso <= 1024).
constexpr int kGwLutMax = 512;        // entries per axis of the per-sample marcher's index tables (larger grids: arithmetic)
template <bool BRICK, bool POW2>
__device__ __forceinline__ void grid_weight_body(const float* __restrict__ sigma, int reso, const float* __restrict__ c2w_all,
                                                 int n_cams, float fx, float fy, int W, int H, const PxoRenderOpts& opt,
                                                 const Vec3& offset, const Vec3& invradius, int* __restrict__ weight_bits,
                                                 uint32_t* __restrict__ s_lut = nullptr, int order = 0) {
  const int tiles_x = (W + 15) / 16, tiles_y = (H + 15) / 16;
  const GwTile gt = gw_tile(blockIdx.x, tiles_x, tiles_y, order);
  if (gt.tile < 0 || gt.cam >= n_cams) return;                 // padding of the supertile order (workgroup-uniform)
  // BRICK && POW2: the bricked index of a voxel is the OR of three per-axis bit patterns; they come from three small LDS
  // tables (s_lut: 3 x kGwLutMax words of workgroup scratch) instead of 13 shift / mask / or instructions per sample.  Filled
  // by the whole workgroup before any thread leaves.
  const bool lut = BRICK && POW2 && s_lut != nullptr && reso <= kGwLutMax;       // workgroup-uniform
  if (lut) {
    const int lbq = 31 - __clz(reso >> 2), lr = lbq + 2;               // reso = 2^lr
    for (int e = threadIdx.x; e < 3 * reso; e += blockDim.x) {
      const int a = e >> lr, x = e & (reso - 1);
      s_lut[a * kGwLutMax + x] = ((uint32_t)(x >> 2) << ((2 - a) * lbq + 6)) | ((uint32_t)(x & 3) << ((2 - a) * 2));
    }
    __syncthreads();
  }
  // 8x8 pixel tiles per 64-thread wave keep the rays of a wave in neighbouring voxels
  const int cam = gt.cam, tile = gt.tile;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int px = (tile % tiles_x) * 16 + (wave & 1) * 8 + (lane & 7);
  const int py = (tile / tiles_x) * 16 + (wave >> 1) * 8 + (lane >> 3);
  if (cam >= n_cams || px >= W || py >= H) return;
  float origin[3], dir[3];
  camera_ray(c2w_all + (int64_t)cam * 12, fx, fy, W, H, px, py, origin, dir);
  TreeRay r;
  to_tree_ray(origin, dir, offset.v, invradius.v, r);
  if (r.tmax < 0.0f || r.tmin > r.tmax) return;
  const float cube = (float)reso;
  const float inv_cube = 1.0f / cube;            // exact when POW2
  const int lb = 31 - __clz(reso >> 2);          // POW2: log2 of bricks per axis
  CellExit cell_exit;
  cell_exit.init(r.invdir, POW2 ? inv_cube : 1.0f);
  // POW2: the grid size is a power of two, so (o + t d) * cube == o * cube + t * (d * cube) and the clamp scales with it, bit
  // for bit (scaling by 2^k commutes with the roundings of the separate mul and add this file is compiled to): the three
  // multiplications by cube leave the loop.  r05: 47 -> 35 vector instructions per sample together with the index tables below.
  const float sc = POW2 ? cube : 1.0f;
  const float so[3] = {r.o[0] * sc, r.o[1] * sc, r.o[2] * sc}, sd[3] = {r.d[0] * sc, r.d[1] * sc, r.d[2] * sc};
  const float chi = (1.0f - 1e-6f) * sc;
  // One sample: the voxel at parameter `tt` and the step to the next sample.  Neither depends on sigma, so the march is
  // SOFTWARE-PIPELINED: the next sample's voxel is located and its sigma load issued before the current sample's sigma is
  // consumed (two loads in flight per ray; the loop was one dependent L2 / HBM round trip per sample).  Same samples, same
  // arithmetic; a ray that stops on `light <= stop_thresh` has read one sigma it did not need.
  auto locate = [&](float tt, int64_t& idx, float& delta_t) {
    float local[3];
    int c[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      // (v_med3_f32: what the compiler makes of clamp_coord's constant bounds; spelled out because `chi` is a register here)
      const float p = POW2 ? __builtin_amdgcn_fmed3f(so[a] + tt * sd[a], 0.0f, chi) : clamp_coord(r.o[a] + tt * r.d[a]) * cube;
      c[a] = (int)p;                             // p >= 0: truncation == floor
      local[a] = __builtin_amdgcn_fractf(p);     // p - floor(p)
    }
    const float s1 = cell_exit(local);
    delta_t = (POW2 ? s1 : s1 / cube) + opt.step_size;
    if (BRICK && POW2) {
      const uint32_t x = (uint32_t)c[0], y = (uint32_t)c[1], z = (uint32_t)c[2];
      if (lut) idx = (uint32_t)(s_lut[x] | s_lut[kGwLutMax + y] | s_lut[2 * kGwLutMax + z]);
      else idx = (uint32_t)(((x >> 2) << (2 * lb + 6)) | ((y >> 2) << (lb + 6)) | ((z >> 2) << 6) | ((x & 3) << 4) | ((y & 3) << 2) | (z & 3));
    } else {
      idx = BRICK ? brick_index(c[0], c[1], c[2], reso >> 2) : ((int64_t)c[0] * reso + c[1]) * reso + c[2];
    }
  };
  float t = r.tmin, light = 1.0f;
  if (!(t < r.tmax)) return;
  int64_t idx;
  float delta_t;
  locate(t, idx, delta_t);
  float sg = sigma[idx];
  for (;;) {
    // the sample after this one (if the march goes on): located and requested now
    const float tn = t + delta_t;
    const bool more = tn > t && tn < r.tmax;     // !(tn > t): step below the resolution of t, stop rather than spin
    // (unconditionally: the position is clamped into the grid, so the address is valid even past tmax, and a straight-line
    // body is what lets the compiler wait for the OLDER of the two loads only -- with the request inside `if (more)` it
    // waited for both at the join and nothing overlapped)
    int64_t idx_n;
    float delta_n;
    locate(tn, idx_n, delta_n);
    const float sg_n = sigma[idx_n];
    if (sg > opt.sigma_thresh) {
      const float att = expf(-(delta_t * r.delta_scale) * sg);
      const float w = light * (1.0f - att);
      light = light * att;
      const int wb = __float_as_int(w);
      if (wb > weight_bits[idx]) atomicMax(weight_bits + idx, wb);     // w >= 0: integer order == float order
      if (light <= opt.stop_thresh) return;
    }
    if (!more) return;
    t = tn; idx = idx_n; delta_t = delta_n; sg = sg_n;
  }
}

// Slab-staged form for power-of-two bricked grids.  A workgroup is a 16x16-pixel tile whose rays sweep the grid together,
// one brick layer ("slab", 4 voxels thick along the dominant axis of the tile's centre ray) at a time: the kWin x kWin bricks
// of that layer around the centre ray are copied to LDS with 16-byte coalesced loads, every ray takes its samples of the slab
// from LDS (sigma) and folds its weights into an LDS copy of the same bricks (ds_max), and the touched weights leave as whole
// bricks (one global atomicMax per voxel and tile instead of a read and possibly an atomic per sample).  Per ray the arithmetic
// and the order of its samples are those of grid_weight_kernel, and max is order-free, so the result is bit-identical.
// Samples outside the window (rare: the window is 24 voxels wide, a tile's footprint at most ~14 + the slab's drift) and rays
// that do not advance along the sweep direction take the global path of the plain kernel.
template <int kWin>
__device__ __forceinline__ void grid_weight_slab_body(const float* __restrict__ sigma, int reso, const float* __restrict__ c2w_all,
                                                      int n_cams, float fx, float fy, int W, int H, const PxoRenderOpts& opt,
                                                      const Vec3& offset, const Vec3& invradius, int* __restrict__ weight_bits,
                                                      float* __restrict__ s_sigma, int* __restrict__ s_w, int order) {
  __shared__ int s_first;
  const int tiles_x = (W + 15) / 16, tiles_y = (H + 15) / 16;
  const GwTile gt = gw_tile(blockIdx.x, tiles_x, tiles_y, order);
  if (gt.tile < 0 || gt.cam >= n_cams) return;                 // padding of the supertile order (workgroup-uniform)
  const int cam = gt.cam, tile = gt.tile;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int tx0 = (tile % tiles_x) * 16, ty0 = (tile / tiles_x) * 16;
  const int px = tx0 + (wave & 1) * 8 + (lane & 7), py = ty0 + (wave >> 1) * 8 + (lane >> 3);
  const float* c2w = c2w_all + (int64_t)cam * 12;
  const int nb = reso >> 2, lb = 31 - __clz(nb);
  const float cube = (float)reso, inv_cube = 1.0f / cube;

  // the tile's centre ray fixes the sweep: axis A, direction sgn, and where the window sits in every slab (workgroup-uniform)
  TreeRay rc;
  {
    float oc[3], dc[3];
    camera_ray(c2w, fx, fy, W, H, min(tx0 + 8, W - 1), min(ty0 + 8, H - 1), oc, dc);
    to_tree_ray(oc, dc, offset.v, invradius.v, rc);
  }
  const float ad0 = fabsf(rc.d[0]), ad1 = fabsf(rc.d[1]), ad2 = fabsf(rc.d[2]);
  const int A = ad0 >= ad1 && ad0 >= ad2 ? 0 : (ad1 >= ad2 ? 1 : 2), B = A == 0 ? 1 : 0, C = A == 2 ? 1 : 2;
  const int sgn = rc.d[A] >= 0.0f ? 1 : -1;
  // bit position of an axis' brick coordinate / in-brick coordinate in the bricked index (brick_index: x slowest)
  const int shA = (2 - A) * lb + 6, shB = (2 - B) * lb + 6, shC = (2 - C) * lb + 6;
  const int lsA = (2 - A) * 2, lsB = (2 - B) * 2, lsC = (2 - C) * 2;

  for (int e = tid; e < kWin * kWin * 64; e += 256) s_w[e] = 0;
  if (tid == 0) s_first = 0x7fffffff;

  // this thread's ray, axes permuted to (A, B, C)
  float po[3] = {0.f, 0.f, 0.f}, pd[3] = {0.f, 0.f, 0.f}, delta_scale = 0.0f, tmax = 0.0f, t = 0.0f, light = 1.0f;
  CellExit cell_exit;
  bool alive = cam < n_cams && px < W && py < H;
  {
    float origin[3], dir[3];
    TreeRay r;
    camera_ray(c2w, fx, fy, W, H, min(px, W - 1), min(py, H - 1), origin, dir);
    to_tree_ray(origin, dir, offset.v, invradius.v, r);
    alive = alive && !(r.tmax < 0.0f || r.tmin > r.tmax) && r.tmin < r.tmax;
    const int perm[3] = {A, B, C};
    float pinv[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      po[i] = perm[i] == 0 ? r.o[0] : (perm[i] == 1 ? r.o[1] : r.o[2]);
      pd[i] = perm[i] == 0 ? r.d[0] : (perm[i] == 1 ? r.d[1] : r.d[2]);
      pinv[i] = perm[i] == 0 ? r.invdir[0] : (perm[i] == 1 ? r.invdir[1] : r.invdir[2]);
    }
    cell_exit.init(pinv, inv_cube);           // the exit distance already divided by the grid size (exact: a power of two)
    delta_scale = r.delta_scale;
    tmax = r.tmax;
    t = r.tmin;
  }
  // the ray in grid units: (o + t d) * cube == o cube + t (d cube), bit for bit, for a power-of-two cube (see grid_weight_body)
  const float so0 = po[0] * cube, so1 = po[1] * cube, so2 = po[2] * cube, sd0 = pd[0] * cube, sd1 = pd[1] * cube, sd2 = pd[2] * cube;
  const float chi = (1.0f - 1e-6f) * cube;
  // the pending sample: cell (permuted axes) and step length at the current t
  int c0 = 0, c1 = 0, c2 = 0;
  float delta_t = 0.0f;
  auto next_sample = [&]() {
    float local[3];
    const float p0 = __builtin_amdgcn_fmed3f(so0 + t * sd0, 0.0f, chi), p1 = __builtin_amdgcn_fmed3f(so1 + t * sd1, 0.0f, chi),
                p2 = __builtin_amdgcn_fmed3f(so2 + t * sd2, 0.0f, chi);
    c0 = (int)p0; c1 = (int)p1; c2 = (int)p2;
    local[0] = __builtin_amdgcn_fractf(p0); local[1] = __builtin_amdgcn_fractf(p1); local[2] = __builtin_amdgcn_fractf(p2);
    delta_t = cell_exit(local) + opt.step_size;
  };
  if (alive) next_sample();
  __syncthreads();
  if (alive) atomicMin(&s_first, (c0 >> 2) * sgn);
  __syncthreads();
  const int first = s_first;
  if (first == 0x7fffffff) return;               // no ray of the tile meets the grid (workgroup-uniform)

  const float cinvA = rc.invdir[A], coA = rc.o[A];
  const float coB = rc.o[B], cdB = rc.d[B], coC = rc.o[C], cdC = rc.d[C];
  for (int key = first;; ++key) {
    const int k = key * sgn;
    if (k < 0 || k >= nb) break;
    // window: kWin x kWin bricks of layer k around the centre ray at the layer's mid-plane
    const float tc = ((float)(4 * k + 2) * inv_cube - coA) * cinvA;
    const int wb0 = (int)floorf((coB + tc * cdB) * cube * 0.25f - 0.5f * (kWin - 1));
    const int wc0 = (int)floorf((coC + tc * cdC) * cube * 0.25f - 0.5f * (kWin - 1));
    for (int e = tid; e < kWin * kWin * 16; e += 256) {
      const int br = e >> 4, q = e & 15;
      const int bi = wb0 + br / kWin, bj = wc0 + br % kWin;
      if ((unsigned)bi < (unsigned)nb && (unsigned)bj < (unsigned)nb) {
        const uint32_t base = ((uint32_t)k << shA) | ((uint32_t)bi << shB) | ((uint32_t)bj << shC);
        *reinterpret_cast<float4*>(s_sigma + br * 64 + q * 4) = *reinterpret_cast<const float4*>(sigma + base + q * 4);
      }
    }
    __syncthreads();
    while (alive && ((c0 >> 2) - k) * sgn <= 0) {
      const int i = (c1 >> 2) - wb0, j = (c2 >> 2) - wc0;
      const bool inwin = (c0 >> 2) == k && (unsigned)i < (unsigned)kWin && (unsigned)j < (unsigned)kWin;
      const uint32_t low = ((uint32_t)(c0 & 3) << lsA) | ((uint32_t)(c1 & 3) << lsB) | ((uint32_t)(c2 & 3) << lsC);
      const int lidx = (i * kWin + j) * 64 + (int)low;
      const uint32_t gidx = ((uint32_t)(c0 >> 2) << shA) | ((uint32_t)(c1 >> 2) << shB) | ((uint32_t)(c2 >> 2) << shC) | low;
      const float sg = inwin ? s_sigma[lidx] : sigma[gidx];
      if (sg > opt.sigma_thresh) {
        const float att = expf(-(delta_t * delta_scale) * sg);
        const float w = light * (1.0f - att);
        light = light * att;
        const int wb = __float_as_int(w);                                  // w >= 0: integer order == float order
        if (inwin) atomicMax(&s_w[lidx], wb);
        else if (wb > weight_bits[gidx]) atomicMax(weight_bits + gidx, wb);
        if (light <= opt.stop_thresh) alive = false;
      }
      const float tn = t + delta_t;
      if (!(tn > t)) alive = false;              // step below the resolution of t: stop rather than spin
      t = tn;
      if (!(t < tmax)) alive = false;
      if (alive) next_sample();
    }
    const int n_alive = __syncthreads_count(alive ? 1 : 0);
    for (int e = tid; e < kWin * kWin * 64; e += 256) {
      const int wv = s_w[e];
      if (wv > 0) {
        const int br = e >> 6;
        const uint32_t g = ((uint32_t)k << shA) | ((uint32_t)(wb0 + br / kWin) << shB) | ((uint32_t)(wc0 + br % kWin) << shC) | (uint32_t)(e & 63);
        if (wv > weight_bits[g]) atomicMax(weight_bits + g, wv);
        s_w[e] = 0;
      }
    }
    if (n_alive == 0) return;
    // (handing the last few live rays of a tile to the per-sample path below -- fewer than 32 .. 192 of 256 -- measured
    // within 1 % on both kinds of scene: not done)
  }
  // rays still alive after the last layer of the sweep (none in practice): the per-sample path of grid_weight_kernel
  while (alive) {
    const uint32_t low = ((uint32_t)(c0 & 3) << lsA) | ((uint32_t)(c1 & 3) << lsB) | ((uint32_t)(c2 & 3) << lsC);
    const uint32_t gidx = ((uint32_t)(c0 >> 2) << shA) | ((uint32_t)(c1 >> 2) << shB) | ((uint32_t)(c2 >> 2) << shC) | low;
    const float sg = sigma[gidx];
    if (sg > opt.sigma_thresh) {
      const float att = expf(-(delta_t * delta_scale) * sg);
      const float w = light * (1.0f - att);
      light = light * att;
      const int wb = __float_as_int(w);
      if (wb > weight_bits[gidx]) atomicMax(weight_bits + gidx, wb);
      if (light <= opt.stop_thresh) break;
    }
    const float tn = t + delta_t;
    if (!(tn > t)) break;
    t = tn;
    if (!(t < tmax)) break;
    next_sample();
  }
}

template <bool BRICK, bool POW2>
__global__ __launch_bounds__(256) void grid_weight_kernel(const float* __restrict__ sigma, int reso,
                                                           const float* __restrict__ c2w_all, int n_cams, float fx, float fy,
                                                           int W, int H, PxoRenderOpts opt, Vec3 offset, Vec3 invradius,
                                                           int* __restrict__ weight_bits) {
  grid_weight_body<BRICK, POW2>(sigma, reso, c2w_all, n_cams, fx, fy, W, H, opt, offset, invradius, weight_bits);
}

// bricked power-of-two grids: either marcher, chosen per launch (workgroup-uniform)
__global__ __launch_bounds__(256) void grid_weight_pow2_kernel(const float* __restrict__ sigma, int reso,
                                                                const float* __restrict__ c2w_all, int n_cams, float fx, float fy,
                                                                int W, int H, PxoRenderOpts opt, Vec3 offset, Vec3 invradius,
                                                                int* __restrict__ weight_bits, GwSelect sel) {
  // one block of workgroup scratch for either marcher: the slab marcher's sigma / weight windows, or the per-sample marcher's
  // index tables (3 x 512 words fit the sigma window)
  constexpr int kWin = 6;
  __shared__ __attribute__((aligned(16))) float s_sigma[kWin * kWin * 64];
  __shared__ int s_w[kWin * kWin * 64];
  static_assert(kWin * kWin * 64 >= 3 * kGwLutMax, "the index tables live in the sigma window");
  if (sel.dense()) grid_weight_slab_body<kWin>(sigma, reso, c2w_all, n_cams, fx, fy, W, H, opt, offset, invradius, weight_bits, s_sigma, s_w, sel.order);
  else grid_weight_body<true, true>(sigma, reso, c2w_all, n_cams, fx, fy, W, H, opt, offset, invradius, weight_bits,
                                    reinterpret_cast<uint32_t*>(s_sigma), sel.order);
}

// ------------------------------------------------------------------------------------------
// octree renderer
// ------------------------------------------------------------------------------------------
constexpr int kRenderThreads = 256;

// lanes per ray: ROW in {4, 8, 16}; the wave carries 64/ROW rays as a WTX x WTY pixel patch
template <int ROW> struct RowGeom {
  static constexpr int kRaysPerWave = 64 / ROW;
  static constexpr int kRaysPerBlock = kRenderThreads / ROW;
  static constexpr int kWTX = kRaysPerWave <= 4 ? 2 : (kRaysPerWave <= 16 ? 4 : 8);
  static constexpr int kWTY = kRaysPerWave / kWTX;
  static constexpr int kMaxLoads = (75 + ROW - 1) / ROW;      // SH25: 75 coefficients
  static constexpr int kMaxGroups = (19 + ROW - 1) / ROW;     // ... as 19 groups of 4 consecutive floats
};
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));   // a leaf row is only 4-byte aligned (49 floats)

template <int ROW>
__device__ __forceinline__ float row_sum(float v) {   // sum over the ROW lanes of a ray, result in every lane
#pragma unroll
  for (int o = 1; o < ROW; o <<= 1) v += __shfl_xor(v, o, ROW);
  return v;
}

// first `basis_dim` SH basis values (the bands of a lower degree are a prefix of the degree-4 list)
__device__ __forceinline__ void sh_basis_dyn(int basis_dim, float x, float y, float z, float* Y) {
  float tmp[25];
  sh_basis<4>(x, y, z, tmp);
#pragma unroll
  for (int i = 0; i < 25; ++i)
    if (i < basis_dim) Y[i] = tmp[i];
}

struct RenderArgs {
  PxoTree tree;
  PxoCamera cam;
  int has_cam;
  const float* origins;
  const float* dirs;
  const float* viewdirs;
  int64_t B;
  PxoRenderOpts opt;
};

// Per-row marching state + the leaf lookup with path reuse.
struct Marcher {
  uint32_t pX, pY, pZ;
  int last_depth;
  int loads;                                   // child-pointer loads of the last find() (read by the counting kernel only)
  bool first;
  int* stack;                                  // LDS, kMaxD + 2 entries of this row

  __device__ __forceinline__ void init(int* s) {
    stack = s;
    first = true;
    last_depth = 0;
    pX = pY = pZ = 0;
    s[0] = 0;
  }
  // finds the leaf containing pos (tree coords, clamped); returns flat cell index node*8+cell
  __device__ __forceinline__ int64_t find(const int32_t* __restrict__ child, const float (&pos)[3], int& depth_out) {
    const uint32_t X = (uint32_t)(pos[0] * (float)(1u << kBits));
    const uint32_t Y = (uint32_t)(pos[1] * (float)(1u << kBits));
    const uint32_t Z = (uint32_t)(pos[2] * (float)(1u << kBits));
    int depth = 0;
    if (!first) {
      const uint32_t diff = (X ^ pX) | (Y ^ pY) | (Z ^ pZ);
      const int common = diff ? (__clz((int)diff) - (32 - kBits)) : kBits;     // leading bit-levels shared
      depth = min(common, last_depth);
    }
    first = false;
    pX = X; pY = Y; pZ = Z;
    int node = stack[depth];
    int cell;
    loads = 0;
    while (true) {
      const int bit = kBits - 1 - depth;
      cell = (int)(((X >> bit) & 1u) << 2 | ((Y >> bit) & 1u) << 1 | ((Z >> bit) & 1u));
      const int skip = child[(int64_t)node * 8 + cell];
      ++loads;
      if (skip == 0 || depth >= kMaxD) break;          // depth bound: a malformed file cannot spin the descent
      node += skip;
      ++depth;
      stack[depth] = node;
    }
    last_depth = depth;
    depth_out = depth;
    return (int64_t)node * 8 + cell;
  }
};

// MODE 0: forward (writes out_rgb).  MODE 1: gradient w.r.t. tree data (two marches per ray).
// VEC (forward only): lane l owns the float groups l, l+ROW, ... (4 consecutive data indices each) instead of the single
// indices l, l+ROW, ...: a leaf's coefficients arrive in ceil(3K/4/ROW) 16-byte loads per lane (SH16, 8 lanes: 2 instead
// of 6 dword loads).  For every SH format 3K+1 is 0 or 1 mod 4, so the last group never leaves the leaf's own row.
// KF != 0 (forward, VEC): channel-aligned ownership.  Per colour channel, lane l owns the float4 groups l, l+ROW, .. of the
// K/4 whole groups and coefficient 4 (K/4) + l of the K % 4 left over, so the basis values a lane needs are the same for
// the three channels: 4 ceil(K/4/ROW) + 1 registers instead of a selector-weighted triple per data element, and a third
// of the FMAs.  KF = K (1, 4, 9, 16, 25) makes K a compile-time constant (SH16 at 4 lanes: one float4 per lane and channel,
// 56 VGPRs - 8 waves per SIMD - against 112 for the index-ordered path); KF = -1 keeps it a run-time value.
template <int MODE, int ROW, bool VEC = false, int KF = 0>
__global__ __launch_bounds__(kRenderThreads) void octree_render_kernel(RenderArgs A, float* __restrict__ out_rgb,
                                                                        const float* __restrict__ fwd_rgb,
                                                                        const float* __restrict__ grad_out,
                                                                        float* __restrict__ grad_data) {
  using G = RowGeom<ROW>;
  constexpr int kRow = ROW, kRaysPerBlock = G::kRaysPerBlock, kMaxLoads = G::kMaxLoads;
  __shared__ int s_stack[kRaysPerBlock][kMaxD + 2];
  __shared__ float s_basis[kRaysPerBlock][25];
  const int row = threadIdx.x / kRow, l = threadIdx.x % kRow;
  int64_t ray;
  float origin[3], dir[3], vdir[3];
  bool active = true;
  if (A.has_cam) {
    // the block's 4 waves form a 2 x 2 arrangement of WTX x WTY pixel patches
    const int W = A.cam.width, H = A.cam.height;
    const int tiles_x = (W + 2 * G::kWTX - 1) / (2 * G::kWTX);
    const int bx = (int)(blockIdx.x % tiles_x), by = (int)(blockIdx.x / tiles_x);
    const int wv = row / G::kRaysPerWave, q = row % G::kRaysPerWave;
    const int px = (bx * 2 + (wv & 1)) * G::kWTX + q % G::kWTX, py = (by * 2 + (wv >> 1)) * G::kWTY + q / G::kWTX;
    active = px < W && py < H;
    ray = (int64_t)py * W + px;
    if (active) {
      camera_ray(A.cam.c2w, A.cam.fx, A.cam.fy, W, H, px, py, origin, dir);
#pragma unroll
      for (int a = 0; a < 3; ++a) vdir[a] = dir[a];
    }
  } else {
    ray = blockIdx.x * (int64_t)kRaysPerBlock + row;
    active = ray < A.B;
    if (active) {
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        origin[a] = A.origins[ray * 3 + a];
        dir[a] = A.dirs[ray * 3 + a];
        vdir[a] = A.viewdirs[ray * 3 + a];
      }
    }
  }
  if (!active) return;                         // whole rows leave together; no block-level barrier below

  const int K = A.tree.basis_dim, D = A.tree.data_dim;
  const float bg = A.opt.background_brightness;
  TreeRay r;
  to_tree_ray(origin, dir, A.tree.offset, A.tree.invradius, r);
  CellExit cell_exit;
  cell_exit.init(r.invdir);
  const bool miss = r.tmax < 0.0f || r.tmin > r.tmax;
  if (MODE == 0 && miss) {
    for (int c = l; c < 3; c += kRow) out_rgb[ray * 3 + c] = bg;
    return;
  }
  if (MODE == 1 && miss) return;

  // per-lane channel ownership: data index l + ROW j -> (channel, SH component)
  if (l == 0) sh_basis_dyn(K, vdir[0], vdir[1], vdir[2], s_basis[row]);
  __builtin_amdgcn_wave_barrier();
  static_assert(!VEC || MODE == 0, "vector loads: forward only (the gradient scatter wants contiguous dword rows)");
  static_assert(KF == 0 || (VEC && MODE == 0), "channel-aligned paths: forward, vector loads");
  const int Kc = KF > 0 ? KF : K;                    // compile-time when the launch is specialised for the tree's format
  constexpr int kChM = ((KF > 0 ? KF / 4 : 6) + ROW - 1) / ROW > 0 ? ((KF > 0 ? KF / 4 : 6) + ROW - 1) / ROW : 1;   // K <= 25: <= 6 whole groups
  const int chG = Kc >> 2, chR = Kc & 3;
  float bk[4 * kChM], bk_r = 0.0f;
  if (KF != 0) {
#pragma unroll
    for (int m = 0; m < kChM; ++m)
#pragma unroll
      for (int e = 0; e < 4; ++e) bk[4 * m + e] = (l + kRow * m) < chG ? s_basis[row][4 * (l + kRow * m) + e] : 0.0f;
    if (l < chR) bk_r = s_basis[row][4 * chG + l];
  }
  constexpr int kSlots = KF ? 1 : (VEC ? G::kMaxGroups * 4 : kMaxLoads);       // data elements owned by a lane (generic paths)
  const int nload = VEC ? ((D - 1 + 3) / 4 + kRow - 1) / kRow : (D - 1 + kRow - 1) / kRow;   // loads per lane in use
  float b0[kSlots], b1[kSlots], b2[kSlots];
#pragma unroll
  for (int j = 0; j < kSlots; ++j) {
    const int idx = VEC ? 4 * (l + kRow * (j >> 2)) + (j & 3) : l + kRow * j;
    float bas = 0.0f;
    int ch = -1;
    if (KF == 0 && (VEC ? (j >> 2) : j) < nload && idx < D - 1) {
      ch = idx / K;
      bas = s_basis[row][idx - ch * K];
    }
    b0[j] = ch == 0 ? bas : 0.0f;
    b1[j] = ch == 1 ? bas : 0.0f;
    b2[j] = ch == 2 ? bas : 0.0f;
  }
  const float* __restrict__ data = A.tree.data;
  const int32_t* __restrict__ child = A.tree.child;

  float g[3] = {0.f, 0.f, 0.f};
  if (MODE == 1) {
#pragma unroll
    for (int c = 0; c < 3; ++c) g[c] = grad_out[ray * 3 + c];
  }
  float accum = 0.0f;                          // MODE 1: sum_c g_c * out_c, then the part behind the sample
  int first_pass = 0;
  if (MODE == 1 && fwd_rgb != nullptr) {       // the caller kept the (exact) forward image: no first march
    accum = (g[0] * fwd_rgb[ray * 3] + g[1] * fwd_rgb[ray * 3 + 1]) + g[2] * fwd_rgb[ray * 3 + 2];
    first_pass = 1;
  }

  // pass 0 composites; in MODE 1 pass 1 re-marches and scatters the gradient
  for (int pass = first_pass; pass <= MODE; ++pass) {
    Marcher mk;
    mk.init(s_stack[row]);
    float t = r.tmin, light = 1.0f;
    float out[3] = {0.f, 0.f, 0.f};
    bool stopped = false;
    // One sample = the leaf at parameter tt and the step to the next sample; neither depends on the leaf's DATA.  The march is
    // software-pipelined on that: a sample's sigma is requested, then the NEXT sample is located (its child-pointer loads
    // travel together with the sigma request), and only then is the current sample shaded -- per sample one dependent L2 round
    // trip less (lookup || sigma -> coefficients instead of lookup -> sigma -> coefficients).  Same samples, same arithmetic.
    auto locate = [&](float tt, int64_t& leaf_o, float& delta_o) {
      float pos[3];
#pragma unroll
      for (int a = 0; a < 3; ++a) pos[a] = clamp_coord(r.o[a] + tt * r.d[a]);
      int depth;
      leaf_o = mk.find(child, pos, depth);
      const float cube = (float)(2u << depth);
      float local[3];
#pragma unroll
      for (int a = 0; a < 3; ++a) local[a] = __builtin_amdgcn_fractf(pos[a] * cube);
      delta_o = cell_exit(local) * inv_cells(depth) + A.opt.step_size;
    };
    int64_t leaf = 0;
    float delta_t = 0.0f;
    bool more = t < r.tmax;
    if (more) locate(t, leaf, delta_t);
    while (more) {
      const float* __restrict__ val = data + leaf * D;
      const float sg = val[D - 1];
      const float tn = t + delta_t;
      more = tn > t && tn < r.tmax;              // !(tn > t): step below the resolution of t, stop rather than spin
      const int64_t leaf_cur = leaf;
      const float delta_cur = delta_t;
      if (more) locate(tn, leaf, delta_t);       // the next sample, while sigma is on its way
      t = tn;
      if (sg > A.opt.sigma_thresh) {
        const float dtw = delta_cur * r.delta_scale;
        const float att = expf(-dtw * sg);
        const float weight = light * (1.0f - att);
        float p0 = 0.f, p1 = 0.f, p2 = 0.f;
        if (KF != 0) {
#pragma unroll
          for (int m = 0; m < kChM; ++m) {
            const int g = l + kRow * m;
            if (g < chG) {
              const f32x4u c0 = *reinterpret_cast<const f32x4u*>(val + 4 * g);
              const f32x4u c1 = *reinterpret_cast<const f32x4u*>(val + Kc + 4 * g);
              const f32x4u c2 = *reinterpret_cast<const f32x4u*>(val + 2 * Kc + 4 * g);
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                p0 += c0[e] * bk[4 * m + e];
                p1 += c1[e] * bk[4 * m + e];
                p2 += c2[e] * bk[4 * m + e];
              }
            }
          }
          if (l < chR) {
            const int k = 4 * chG + l;
            p0 += val[k] * bk_r;
            p1 += val[Kc + k] * bk_r;
            p2 += val[2 * Kc + k] * bk_r;
          }
        } else if (VEC) {
#pragma unroll
          for (int j = 0; j < G::kMaxGroups; ++j) {
            const int g4 = 4 * (l + kRow * j);
            if (j < nload && g4 < D - 1) {
              const f32x4u v4 = *reinterpret_cast<const f32x4u*>(val + g4);
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float v = g4 + e < D - 1 ? v4[e] : 0.0f;          // the row's last float is sigma, not a coefficient
                p0 += v * b0[4 * j + e];
                p1 += v * b1[4 * j + e];
                p2 += v * b2[4 * j + e];
              }
            }
          }
        } else {
#pragma unroll
          for (int j = 0; j < kMaxLoads; ++j) {
            if (j < nload) {
              const int idx = l + kRow * j;
              const float v = idx < D - 1 ? val[idx] : 0.0f;
              p0 += v * b0[j];
              p1 += v * b1[j];
              p2 += v * b2[j];
            }
          }
        }
        p0 = row_sum<ROW>(p0);
        p1 = row_sum<ROW>(p1);
        p2 = row_sum<ROW>(p2);
        const float c0 = 1.0f / (1.0f + expf(-p0)), c1 = 1.0f / (1.0f + expf(-p1)), c2 = 1.0f / (1.0f + expf(-p2));
        if (pass == 0) {
          out[0] += weight * c0;
          out[1] += weight * c1;
          out[2] += weight * c2;
          light = light * att;
          if (MODE == 0 && light <= A.opt.stop_thresh) {
            const float scale = 1.0f / (1.0f - light);
            out[0] *= scale; out[1] *= scale; out[2] *= scale;
            stopped = true;
            break;
          }
        } else {
          const float total = (g[0] * c0 + g[1] * c1) + g[2] * c2;
          const float d0 = weight * g[0] * c0 * (1.0f - c0);
          const float d1 = weight * g[1] * c1 * (1.0f - c1);
          const float d2 = weight * g[2] * c2 * (1.0f - c2);
          float* __restrict__ gv = grad_data + leaf_cur * D;
#pragma unroll
          for (int j = 0; j < kMaxLoads; ++j) {
            if (j < nload) {
              const int idx = l + kRow * j;
              if (idx < D - 1) unsafeAtomicAdd(gv + idx, (b0[j] * d0 + b1[j] * d1) + b2[j] * d2);
            }
          }
          light = light * att;
          accum -= weight * total;
          if (l == 0) unsafeAtomicAdd(gv + D - 1, dtw * (total * light - accum));
        }
      }
    }
    if (pass == 0) {
      if (MODE == 0) {
        if (!stopped) {
          out[0] += light * bg; out[1] += light * bg; out[2] += light * bg;
        }
        for (int c = l; c < 3; c += kRow) out_rgb[ray * 3 + c] = c == 0 ? out[0] : (c == 1 ? out[1] : out[2]);
      } else {
        accum = (g[0] * (out[0] + light * bg) + g[1] * (out[1] + light * bg)) + g[2] * (out[2] + light * bg);
      }
    }
  }
}

// Backward with a 4-lane march and a 16-lane scatter.  The march is the forward kernel's: 16 rays per wave,
// channel-aligned coefficient reads, few registers - the phase that is latency-bound.  The scatter wants the opposite shape
// (the gradient of a sample is a 3K-float row: 64-byte atomic rows per instruction), so after every march step the wave
// re-deals itself as 4 rays x 16 lanes, four times: lane (q, j) of deal g fetches the step's row-uniform results of ray
// 4g + q from that ray's lanes (shfl) and adds basis_k(ray) * d_channel at the data indices j, j + 16, .. - the same 64-byte rows per
// instruction as the 16-lane kernel, with the march running at four times its rays per wave.  Rays that have finished stay in
// the loop (idle) until the whole wave is done, since every lane takes part in every deal.
// UPD (WC > 0): how a sample's gradient row reaches the wave's cache.
//   0  one sample at a time, the whole wave on its 3K+1-float row (15 of 64 lanes idle for SH16, ~27 vector instructions per
//      sample, ~12 samples per march step of the wave's 16 rays: the update phase issued twice the instructions of the march);
//   1  every ray's own 4 lanes add its sample's row -- 13 (SH16) / 19 (SH25) LDS float atomics per lane, all 16 rays at once --
//      when the slot already holds the ray's leaf (a hit); rays that miss elect ONE winner per slot through LDS, the winners' old
//      rows leave as before (one contiguous row of global atomics per row, the whole wave on it) and the winners write their new
//      rows themselves; rays that lost an election go round again and usually hit the row the winner has just installed (the
//      neighbouring pixel entered the same leaf in the same step).  A wave's LDS operations execute in program order, so the
//      hits of a round land before the evictions read the rows and the installs after.  Measured (r05c): 4.27 / 4.28 ms against
//      4.26 / 4.29 ms for form 0 -- no gain (the kernel is bound by its evicted rows, not by issue): form 0 stays the default.
template <int KF, int WC, int UPD = 0>
__global__ __launch_bounds__(kRenderThreads) void octree_render_bwd4_kernel(RenderArgs A, const float* __restrict__ fwd_rgb,
                                                                             const float* __restrict__ grad_out,
                                                                             float* __restrict__ grad_data) {
  using G = RowGeom<4>;
  constexpr int kRow = 4, kRaysPerBlock = G::kRaysPerBlock;
  constexpr int kDmax = 3 * (KF > 0 ? KF : 25) + 1, kRM = (kDmax + 63) / 64, kWCn = WC > 0 ? WC : 1;
  constexpr int kPM = (kDmax + kRow - 1) / kRow;                 // row elements per lane of a ray (UPD 1): l, l + 4, ...
  __shared__ int s_stack[kRaysPerBlock][kMaxD + 2];
  __shared__ float s_basis[kRaysPerBlock][25];
  __shared__ float s_rows[kRenderThreads / 64][kWCn][WC > 0 ? kDmax : 1];
  __shared__ int s_tag[kRenderThreads / 64][UPD ? kWCn : 1];     // UPD 1: the leaf whose row sits in slot i (-1: empty)
  __shared__ int s_own[kRenderThreads / 64][UPD ? kWCn : 1];     //        the ray that won the slot in this round
  static_assert(WC <= 64, "one tag per lane");
  static_assert(UPD == 0 || WC > 0, "the ray-parallel update needs the cache");
  const int row = threadIdx.x / kRow, l = threadIdx.x % kRow, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int tagv = -1;                                 // lane i: the leaf whose row sits in slot i of this wave's cache (-1: empty)
  int64_t ray = 0;
  float origin[3] = {0.f, 0.f, 0.f}, dir[3] = {0.f, 0.f, 1.f}, vdir[3] = {0.f, 0.f, 1.f};
  bool active = true;
  if (A.has_cam) {
    const int W = A.cam.width, H = A.cam.height;
    const int tiles_x = (W + 2 * G::kWTX - 1) / (2 * G::kWTX);
    const int bx = (int)(blockIdx.x % tiles_x), by = (int)(blockIdx.x / tiles_x);
    const int wv = row / G::kRaysPerWave, q = row % G::kRaysPerWave;
    const int px = (bx * 2 + (wv & 1)) * G::kWTX + q % G::kWTX, py = (by * 2 + (wv >> 1)) * G::kWTY + q / G::kWTX;
    active = px < W && py < H;
    ray = (int64_t)py * W + px;
    if (active) {
      camera_ray(A.cam.c2w, A.cam.fx, A.cam.fy, W, H, px, py, origin, dir);
#pragma unroll
      for (int a = 0; a < 3; ++a) vdir[a] = dir[a];
    }
  } else {
    ray = blockIdx.x * (int64_t)kRaysPerBlock + row;
    active = ray < A.B;
    if (active) {
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        origin[a] = A.origins[ray * 3 + a];
        dir[a] = A.dirs[ray * 3 + a];
        vdir[a] = A.viewdirs[ray * 3 + a];
      }
    }
  }
  const int K = A.tree.basis_dim, D = A.tree.data_dim;
  const int Kc = KF > 0 ? KF : K;
  const float bg = A.opt.background_brightness;
  TreeRay r;
  to_tree_ray(origin, dir, A.tree.offset, A.tree.invradius, r);
  CellExit cell_exit;
  cell_exit.init(r.invdir);
  const bool alive = active && !(r.tmax < 0.0f || r.tmin > r.tmax);

  if (l == 0) {
    if (alive) sh_basis_dyn(K, vdir[0], vdir[1], vdir[2], s_basis[row]);
    else
      for (int i = 0; i < 25; ++i) s_basis[row][i] = 0.0f;
  }
  __builtin_amdgcn_wave_barrier();
  // march side: channel-aligned ownership (see octree_render_kernel, KF)
  constexpr int kChM = ((KF > 0 ? KF / 4 : 6) + kRow - 1) / kRow > 0 ? ((KF > 0 ? KF / 4 : 6) + kRow - 1) / kRow : 1;
  const int chG = Kc >> 2, chR = Kc & 3;
  float bk[4 * kChM], bk_r = 0.0f;
#pragma unroll
  for (int m = 0; m < kChM; ++m)
#pragma unroll
    for (int e = 0; e < 4; ++e) bk[4 * m + e] = (l + kRow * m) < chG ? s_basis[row][4 * (l + kRow * m) + e] : 0.0f;
  if (l < chR) bk_r = s_basis[row][4 * chG + l];
  // scatter side: in deal g this lane serves ray 4 g + (lane >> 4) of the wave, data indices j, j + 16, .. of the leaf row
  // (index-ordered, so every instruction adds 64 contiguous bytes per ray whatever K is)
  constexpr int kSM = (3 * (KF > 0 ? KF : 25) + 15) / 16;
  const int j = lane & 15, wave_row0 = (row / G::kRaysPerWave) * G::kRaysPerWave;
  float sb[4][kSM];
  int sch[kSM];
#pragma unroll
  for (int m = 0; m < kSM; ++m) {
    const int idx = j + 16 * m;
    sch[m] = idx < 3 * Kc ? idx / Kc : -1;
#pragma unroll
    for (int gq = 0; gq < 4; ++gq)
      sb[gq][m] = sch[m] >= 0 ? s_basis[wave_row0 + 4 * gq + (lane >> 4)][idx - sch[m] * Kc] : 0.0f;
  }
  // write-combining side: the wave's lane owns the data indices lane, lane + 64 of whatever row is being updated
  int wch[kRM], wof[kRM];
#pragma unroll
  for (int m = 0; m < kRM; ++m) {
    const int idx = lane + 64 * m;
    wch[m] = idx < 3 * Kc ? idx / Kc : 3;
    wof[m] = idx < 3 * Kc ? idx - wch[m] * Kc : 0;
  }
  // ray-parallel side (UPD 1): this lane owns the data indices l, l + 4, .. of its OWN ray's sample row: basis value and channel
  // (3 = the sigma entry, 4 = past the row) per owned index
  float pb[UPD ? kPM : 1];
  int pch[UPD ? kPM : 1];
  if (UPD) {
#pragma unroll
    for (int m = 0; m < kPM; ++m) {
      const int idx = l + kRow * m;
      pch[m] = idx < 3 * Kc ? idx / Kc : (idx == D - 1 ? 3 : 4);
      pb[m] = idx < 3 * Kc ? s_basis[row][idx - pch[m] * Kc] : (idx == D - 1 ? 1.0f : 0.0f);
    }
    if (lane < kWCn) s_tag[wave][lane] = -1;
  }
  const int ray_in_wave = lane >> 2;
  const float* __restrict__ data = A.tree.data;
  const int32_t* __restrict__ child = A.tree.child;

  float g[3] = {0.f, 0.f, 0.f};
  if (alive) {
#pragma unroll
    for (int c = 0; c < 3; ++c) g[c] = grad_out[ray * 3 + c];
  }
  float accum = 0.0f;
  int first_pass = 0;
  if (fwd_rgb != nullptr) {                    // the caller kept the (exact) forward image: no first march
    if (alive) accum = (g[0] * fwd_rgb[ray * 3] + g[1] * fwd_rgb[ray * 3 + 1]) + g[2] * fwd_rgb[ray * 3 + 2];
    first_pass = 1;
  }
  for (int pass = first_pass; pass <= 1; ++pass) {
    Marcher mk;
    mk.init(s_stack[row]);
    float t = r.tmin, light = 1.0f;
    float out[3] = {0.f, 0.f, 0.f};
    bool running = alive && t < r.tmax;
    while (__builtin_amdgcn_ballot_w64(running) != 0) {
      bool has = false;
      int leaf_i = 0;
      float e0 = 0.f, e1 = 0.f, e2 = 0.f, es = 0.f;
      if (running) {
        float pos[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) pos[a] = clamp_coord(r.o[a] + t * r.d[a]);
        int depth;
        const int64_t leaf = mk.find(child, pos, depth);
        const float cube = (float)(2u << depth);
        float local[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) local[a] = __builtin_amdgcn_fractf(pos[a] * cube);
        const float delta_t = cell_exit(local) * inv_cells(depth) + A.opt.step_size;
        const float* __restrict__ val = data + leaf * D;
        const float sg = val[D - 1];
        if (sg > A.opt.sigma_thresh) {
          const float dtw = delta_t * r.delta_scale;
          const float att = expf(-dtw * sg);
          const float weight = light * (1.0f - att);
          float p0 = 0.f, p1 = 0.f, p2 = 0.f;
#pragma unroll
          for (int m = 0; m < kChM; ++m) {
            const int gi = l + kRow * m;
            if (gi < chG) {
              const f32x4u c0 = *reinterpret_cast<const f32x4u*>(val + 4 * gi);
              const f32x4u c1 = *reinterpret_cast<const f32x4u*>(val + Kc + 4 * gi);
              const f32x4u c2 = *reinterpret_cast<const f32x4u*>(val + 2 * Kc + 4 * gi);
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                p0 += c0[e] * bk[4 * m + e];
                p1 += c1[e] * bk[4 * m + e];
                p2 += c2[e] * bk[4 * m + e];
              }
            }
          }
          if (l < chR) {
            const int k = 4 * chG + l;
            p0 += val[k] * bk_r;
            p1 += val[Kc + k] * bk_r;
            p2 += val[2 * Kc + k] * bk_r;
          }
          p0 = row_sum<kRow>(p0);
          p1 = row_sum<kRow>(p1);
          p2 = row_sum<kRow>(p2);
          const float c0 = 1.0f / (1.0f + expf(-p0)), c1 = 1.0f / (1.0f + expf(-p1)), c2 = 1.0f / (1.0f + expf(-p2));
          if (pass == 0) {
            out[0] += weight * c0;
            out[1] += weight * c1;
            out[2] += weight * c2;
            light = light * att;
          } else {
            const float total = (g[0] * c0 + g[1] * c1) + g[2] * c2;
            e0 = weight * g[0] * c0 * (1.0f - c0);
            e1 = weight * g[1] * c1 * (1.0f - c1);
            e2 = weight * g[2] * c2 * (1.0f - c2);
            light = light * att;
            accum -= weight * total;
            es = dtw * (total * light - accum);
            leaf_i = (int)leaf;                    // n_internal < 2^28: node * 8 + cell fits 31 bits
            has = true;
          }
        }
        const float tn = t + delta_t;
        running = tn > t && tn < r.tmax;           // !(tn > t): step below the resolution of t, stop rather than spin
        t = tn;
      }
      if (pass == 1 && WC > 0 && UPD == 1) {
        float vv[kPM];
#pragma unroll
        for (int m = 0; m < kPM; ++m)
          vv[m] = pb[m] * (pch[m] == 0 ? e0 : (pch[m] == 1 ? e1 : (pch[m] == 2 ? e2 : es)));
        const int slot = (int)(((uint32_t)leaf_i * 2654435761u) >> 16) & (kWCn - 1);
        float* const rowp = s_rows[wave][slot] + l;
        bool pend = has;
        while (__builtin_amdgcn_ballot_w64(pend) != 0) {
          const int tag = pend ? s_tag[wave][slot] : -2;
          if (pend && tag == leaf_i) {                          // hit: this ray's 4 lanes add its row
#pragma unroll
            for (int m = 0; m < kPM; ++m)
              if (pch[m] < 4) __hip_atomic_fetch_add(rowp + kRow * m, vv[m], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            pend = false;
          }
          if (pend && l == 0) s_own[wave][slot] = ray_in_wave;   // the rays that missed: one winner per slot
          const bool win = pend && s_own[wave][slot] == ray_in_wave;
          // the winners' old rows leave, each as ONE contiguous row of atomics with the whole wave on it
          uint64_t em = __builtin_amdgcn_ballot_w64(win && l == 0 && tag >= 0);
          while (em) {
            const int src = __builtin_ctzll(em);
            em &= em - 1;
            const int sl = __builtin_amdgcn_readlane(slot, src);
            const int tg = __builtin_amdgcn_readlane(tag, src);
            const float* rp = s_rows[wave][sl];
#pragma unroll
            for (int m = 0; m < kRM; ++m) {
              const int idx = lane + 64 * m;
              if (idx < D) unsafeAtomicAdd(grad_data + (int64_t)tg * D + idx, rp[idx]);
            }
          }
          if (win) {                                            // ... and the winners start their new rows
#pragma unroll
            for (int m = 0; m < kPM; ++m)
              if (pch[m] < 4) rowp[kRow * m] = vv[m];
            if (l == 0) s_tag[wave][slot] = leaf_i;
            pend = false;
          }
        }
      }
      if (pass == 1 && WC > 0 && UPD == 0) {
        // one sample at a time, the whole wave on its row: hit -> add in LDS; miss -> the evicted row leaves as ONE
        // contiguous row of atomics, the new row starts from this sample
        uint64_t hm = __builtin_amdgcn_ballot_w64(has && l == 0);
        while (hm) {
          const int src = __builtin_ctzll(hm);
          hm &= hm - 1;
          const int lf = __builtin_amdgcn_readlane(leaf_i, src);
          const float q0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(e0), src));
          const float q1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(e1), src));
          const float q2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(e2), src));
          const float qs = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(es), src));
          const int slot = (int)(((uint32_t)lf * 2654435761u) >> 16) & (kWCn - 1);
          const int tag = __builtin_amdgcn_readlane(tagv, slot);
          float* rowp = s_rows[wave][slot];
          const float* bas = s_basis[wave_row0 + (src >> 2)];
#pragma unroll
          for (int m = 0; m < kRM; ++m) {
            const int idx = lane + 64 * m;
            if (idx < D) {
              const float v = wch[m] < 3 ? bas[wof[m]] * (wch[m] == 0 ? q0 : (wch[m] == 1 ? q1 : q2)) : qs;
              if (tag == lf) {
                rowp[idx] += v;          // (as ds_add_f32, one LDS instruction instead of read-add-write: measured 1.9 % SLOWER,
                                         //  4.82 vs 4.73 ms per image, round 4: the LDS atomic unit is the slower path)
              } else {
                if (tag >= 0) unsafeAtomicAdd(grad_data + (int64_t)tag * D + idx, rowp[idx]);
                rowp[idx] = v;
              }
            }
          }
          tagv = lane == slot ? lf : tagv;
        }
      }
      if (pass == 1 && WC == 0) {
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const int src = 4 * (4 * gq + (lane >> 4));            // first lane of the served ray's row
          const int h = __shfl((int)has, src);
          if (__builtin_amdgcn_ballot_w64(h != 0) == 0) continue;   // no sample in this deal (wave-uniform)
          const int lf = __shfl(leaf_i, src);
          const float q0 = __shfl(e0, src), q1 = __shfl(e1, src), q2 = __shfl(e2, src), qs = __shfl(es, src);
          if (h) {
            float* __restrict__ gv = grad_data + (int64_t)lf * D;
#pragma unroll
            for (int m = 0; m < kSM; ++m)
              if (sch[m] >= 0)
                unsafeAtomicAdd(gv + j + 16 * m, sb[gq][m] * (sch[m] == 0 ? q0 : (sch[m] == 1 ? q1 : q2)));
            if (j == 0) unsafeAtomicAdd(gv + D - 1, qs);
          }
        }
      }
    }
    if (pass == 0) accum = (g[0] * (out[0] + light * bg) + g[1] * (out[1] + light * bg)) + g[2] * (out[2] + light * bg);
  }
  if (WC > 0) {                                  // rows still held by the wave
    for (int sl = 0; sl < WC; ++sl) {
      const int tag = UPD ? __builtin_amdgcn_readfirstlane(s_tag[wave][sl]) : __builtin_amdgcn_readlane(tagv, sl);
      if (tag < 0) continue;
#pragma unroll
      for (int m = 0; m < kRM; ++m) {
        const int idx = lane + 64 * m;
        if (idx < D) unsafeAtomicAdd(grad_data + (int64_t)tag * D + idx, s_rows[wave][sl][idx]);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// work counters (roofline pass, scripts/octree_bench.py): the marchers' own ray set-up, leaf lookup, step rule and
// early stop -- same arithmetic, so the same sample sequence -- counting instead of shading.  One thread per ray.
//   counts[0] rays that enter the volume      counts[2] samples above sigma_thresh (one full row / one weight update each)
//   counts[1] samples (one sigma read each)   counts[3] child-pointer loads of the leaf lookups (tree marcher only)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void flush_counts(unsigned long long (&c)[4], unsigned long long* __restrict__ counts) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    unsigned long long v = c[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(counts + i, v);
  }
}

__global__ __launch_bounds__(256) void octree_count_kernel(RenderArgs A, unsigned long long* __restrict__ counts,
                                                           uint8_t* __restrict__ leaf_seen) {
  __shared__ int s_stack[256][kMaxD + 2];
  unsigned long long c[4] = {0, 0, 0, 0};
  const int W = A.cam.width, H = A.cam.height;
  const int64_t ray = blockIdx.x * (int64_t)256 + threadIdx.x;
  if (ray < (int64_t)W * H) {
    float origin[3], dir[3];
    camera_ray(A.cam.c2w, A.cam.fx, A.cam.fy, W, H, (int)(ray % W), (int)(ray / W), origin, dir);
    TreeRay r;
    to_tree_ray(origin, dir, A.tree.offset, A.tree.invradius, r);
    CellExit cell_exit;
    cell_exit.init(r.invdir);
    if (!(r.tmax < 0.0f || r.tmin > r.tmax)) {
      c[0] = 1;
      const int D = A.tree.data_dim;
      const float* __restrict__ data = A.tree.data;
      const int32_t* __restrict__ child = A.tree.child;
      Marcher mk;
      mk.init(s_stack[threadIdx.x]);
      float t = r.tmin, light = 1.0f;
      while (t < r.tmax) {
        float pos[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) pos[a] = clamp_coord(r.o[a] + t * r.d[a]);
        int depth;
        const int64_t leaf = mk.find(child, pos, depth);
        c[3] += (unsigned long long)mk.loads;
        const float cube = (float)(2u << depth);
        float local[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) local[a] = __builtin_amdgcn_fractf(pos[a] * cube);
        const float delta_t = cell_exit(local) * inv_cells(depth) + A.opt.step_size;
        const float sg = data[leaf * D + D - 1];
        c[1] += 1;
        if (sg > A.opt.sigma_thresh) {
          c[2] += 1;
          if (leaf_seen) leaf_seen[leaf] = 1;
          light = light * expf(-(delta_t * r.delta_scale) * sg);
          if (light <= A.opt.stop_thresh) break;
        }
        const float tn = t + delta_t;
        if (!(tn > t)) break;
        t = tn;
      }
    }
  }
  flush_counts(c, counts);
}

__global__ __launch_bounds__(256) void grid_count_kernel(const float* __restrict__ sigma, int reso, const float* __restrict__ c2w_all,
                                                         int n_cams, float fx, float fy, int W, int H, PxoRenderOpts opt,
                                                         Vec3 offset, Vec3 invradius, unsigned long long* __restrict__ counts,
                                                         uint8_t* __restrict__ voxel_seen) {
  unsigned long long c[4] = {0, 0, 0, 0};
  const int64_t id = blockIdx.x * (int64_t)256 + threadIdx.x;
  const int64_t hw = (int64_t)W * H;
  if (id < hw * n_cams) {
    const int cam = (int)(id / hw);
    const int64_t pix = id % hw;
    float origin[3], dir[3];
    camera_ray(c2w_all + (int64_t)cam * 12, fx, fy, W, H, (int)(pix % W), (int)(pix / W), origin, dir);
    TreeRay r;
    to_tree_ray(origin, dir, offset.v, invradius.v, r);
    if (!(r.tmax < 0.0f || r.tmin > r.tmax)) {
      c[0] = 1;
      const float cube = (float)reso;
      CellExit cell_exit;
      cell_exit.init(r.invdir);
      float t = r.tmin, light = 1.0f;
      while (t < r.tmax) {
        float local[3];
        int cc[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          const float p = clamp_coord(r.o[a] + t * r.d[a]) * cube;
          cc[a] = (int)p;
          local[a] = __builtin_amdgcn_fractf(p);
        }
        const float delta_t = cell_exit(local) / cube + opt.step_size;     // reso a power of two: the exact quotient either way
        const int64_t idx = ((int64_t)cc[0] * reso + cc[1]) * reso + cc[2];
        const float sg = sigma[idx];
        c[1] += 1;
        if (sg > opt.sigma_thresh) {
          c[2] += 1;
          if (voxel_seen) voxel_seen[idx] = 1;
          light = light * expf(-(delta_t * r.delta_scale) * sg);
          if (light <= opt.stop_thresh) break;
        }
        const float tn = t + delta_t;
        if (!(tn > t)) break;
        t = tn;
      }
    }
  }
  flush_counts(c, counts);
}

// ------------------------------------------------------------------------------------------
// image loss and SGD
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void image_mse_kernel(const float* __restrict__ im, const float* __restrict__ gt, int64_t n,
                                                         float* __restrict__ grad, float* __restrict__ sse) {
  __shared__ float s_part[4];
  float acc = 0.0f;
  const float scale = 2.0f / (float)n;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float v = im[i];
    const float c = fminf(fmaxf(v, 0.0f), 1.0f);
    const float e = c - gt[i];
    acc += e * e;
    if (grad) grad[i] = (v >= 0.0f && v <= 1.0f) ? scale * e : 0.0f;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(sse, (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]));
}

__global__ void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf, int64_t n, float lr,
                           float mu, int nesterov, int first) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  float gi = g[i];
  if (mu != 0.0f) {
    const float b = first ? gi : mu * buf[i] + gi;
    buf[i] = b;
    gi = nesterov ? gi + mu * b : b;
  }
  p[i] = p[i] - lr * gi;
}

}  // namespace pxo

// ==========================================================================================
// C ABI
// ==========================================================================================
using namespace pxo;

static inline int64_t blocks_for(int64_t n, int threads) { return (n + threads - 1) / threads; }

extern "C" {

int pxo_threshold_mask(const float* value, int64_t n, float thresh, uint8_t* mask, void* stream) {
  PXO_REQUIRE(n >= 0, "pxo_threshold_mask: n < 0");
  if (n == 0) return PXO_OK;
  PXO_REQUIRE(value && mask, "pxo_threshold_mask: null pointer");
  hipLaunchKernelGGL(threshold_mask_kernel, dim3((unsigned)blocks_for(n, 256)), dim3(256), 0, (hipStream_t)stream, value, n,
                     thresh, mask);
  return check_launch("threshold_mask");
}

int pxo_tree_workspace_bytes(int depth, size_t* bytes) {
  PXO_REQUIRE(depth >= 1 && depth <= kMaxD, "pxo_tree_workspace_bytes: depth %d outside [1,%d]", depth, kMaxD);
  PXO_REQUIRE(bytes, "pxo_tree_workspace_bytes: null pointer");
  *bytes = (size_t)tree_ws(depth).total;
  return PXO_OK;
}

static int scan_level(const uint8_t* occ, int64_t n, uint32_t* bsum, int32_t* rank, int64_t* count, hipStream_t s) {
  const int64_t nb = blocks_for(n, kScanElems);
  hipLaunchKernelGGL(scan_blocksum_kernel, dim3((unsigned)nb), dim3(256), 0, s, occ, n, bsum);
  hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(256), 0, s, bsum, nb, count);
  hipLaunchKernelGGL(scan_final_kernel, dim3((unsigned)nb), dim3(256), 0, s, occ, n, (const uint32_t*)bsum, rank);
  return check_launch("tree scan");
}

int pxo_tree_count_nodes(const uint8_t* mask, int depth, void* ws, size_t ws_bytes, int64_t* level_nodes, void* stream) {
  PXO_REQUIRE(depth >= 1 && depth <= kMaxD, "pxo_tree_count_nodes: depth %d outside [1,%d]", depth, kMaxD);
  PXO_REQUIRE(mask && ws && level_nodes, "pxo_tree_count_nodes: null pointer");
  const TreeWs w = tree_ws(depth);
  if (ws_bytes < (size_t)w.total) {
    set_error("pxo_tree_count_nodes: workspace %zu < %lld", ws_bytes, (long long)w.total);
    return PXO_ERR_WORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  char* base = (char*)ws;
  uint8_t* occ_d = (uint8_t*)(base + w.occ_off[depth]);
  hipLaunchKernelGGL(pyramid_base_kernel, dim3((unsigned)blocks_for(pow8(depth), 256)), dim3(256), 0, s, mask, depth, occ_d);
  for (int d = depth - 1; d >= 1; --d) {
    hipLaunchKernelGGL(pyramid_up_kernel, dim3((unsigned)blocks_for(pow8(d), 256)), dim3(256), 0, s,
                       (const uint8_t*)(base + w.occ_off[d + 1]), pow8(d), (uint8_t*)(base + w.occ_off[d]));
  }
  int64_t* counts = (int64_t*)(base + w.count_off);
  for (int d = 1; d <= depth; ++d) {
    const int rc = scan_level((const uint8_t*)(base + w.occ_off[d]), pow8(d), (uint32_t*)(base + w.bsum_off),
                              (int32_t*)(base + w.rank_off[d]), counts + d, s);
    if (rc) return rc;
  }
  if (hipMemcpyAsync(level_nodes + 1, counts + 1, sizeof(int64_t) * depth, hipMemcpyDeviceToHost, s) != hipSuccess ||
      hipStreamSynchronize(s) != hipSuccess) {
    set_error("pxo_tree_count_nodes: %s", hipGetErrorString(hipGetLastError()));
    return PXO_ERR_HIP;
  }
  level_nodes[0] = 1;
  return PXO_OK;
}

int pxo_tree_build(const void* ws, size_t ws_bytes, int depth, const int64_t* level_nodes, int32_t* child,
                   int32_t* parent_depth, void* stream) {
  PXO_REQUIRE(depth >= 1 && depth <= kMaxD, "pxo_tree_build: depth %d outside [1,%d]", depth, kMaxD);
  PXO_REQUIRE(ws && level_nodes && child && parent_depth, "pxo_tree_build: null pointer");
  const TreeWs w = tree_ws(depth);
  if (ws_bytes < (size_t)w.total) {
    set_error("pxo_tree_build: workspace %zu < %lld", ws_bytes, (long long)w.total);
    return PXO_ERR_WORKSPACE;
  }
  LevelPtrs lv{};
  const char* base = (const char*)ws;
  int64_t start = 0;
  for (int d = 0; d <= depth; ++d) {
    lv.start[d] = start;
    start += level_nodes[d];
    if (d >= 1) {
      lv.occ[d] = (const uint8_t*)(base + w.occ_off[d]);
      lv.rank[d] = (const int32_t*)(base + w.rank_off[d]);
    }
  }
  PXO_REQUIRE(start < ((int64_t)1 << 28), "pxo_tree_build: %lld nodes exceed the int32 packed-index range", (long long)start);
  for (int d = 0; d <= depth; ++d) {
    if (level_nodes[d] == 0) break;
    hipLaunchKernelGGL(tree_emit_kernel, dim3((unsigned)blocks_for(pow8(d), 256)), dim3(256), 0, (hipStream_t)stream, lv, d,
                       depth, child, parent_depth);
  }
  return check_launch("tree_emit");
}

int pxo_tree_sample_cells(const int32_t* parent_depth, int64_t node0, int64_t n_nodes, int S, const float* u,
                          const float offset[3], const float invradius[3], float* points, void* stream) {
  PXO_REQUIRE(n_nodes >= 0 && node0 >= 0 && S >= 1, "pxo_tree_sample_cells: bad sizes");
  if (n_nodes == 0) return PXO_OK;
  PXO_REQUIRE(parent_depth && u && offset && invradius && points, "pxo_tree_sample_cells: null pointer");
  const int64_t n = n_nodes * 8 * S;
  hipLaunchKernelGGL(tree_sample_cells_kernel, dim3((unsigned)blocks_for(n, 256)), dim3(256), 0, (hipStream_t)stream,
                     parent_depth, node0, n_nodes, S, u, offset[0], offset[1], offset[2], invradius[0], invradius[1],
                     invradius[2], points);
  return check_launch("tree_sample_cells");
}

int pxo_tree_sample_leaves(const int32_t* parent_depth, const int64_t* packed, int64_t n_cells, int S, const float* u,
                           const float offset[3], const float invradius[3], float* points, void* stream) {
  PXO_REQUIRE(n_cells >= 0 && S >= 1, "pxo_tree_sample_leaves: bad sizes");
  if (n_cells == 0) return PXO_OK;
  PXO_REQUIRE(parent_depth && packed && u && offset && invradius && points, "pxo_tree_sample_leaves: null pointer");
  const int64_t n = n_cells * S;
  hipLaunchKernelGGL(tree_sample_leaves_kernel, dim3((unsigned)blocks_for(n, 256)), dim3(256), 0, (hipStream_t)stream,
                     parent_depth, packed, n_cells, S, u, offset[0], offset[1], offset[2], invradius[0], invradius[1],
                     invradius[2], points);
  return check_launch("tree_sample_leaves");
}

int pxo_tree_query(const int32_t* child, const float* points, int64_t n, const float offset[3], const float invradius[3],
                   int64_t* packed, void* stream) {
  PXO_REQUIRE(n >= 0, "pxo_tree_query: n < 0");
  if (n == 0) return PXO_OK;
  PXO_REQUIRE(child && points && offset && invradius && packed, "pxo_tree_query: null pointer");
  hipLaunchKernelGGL(tree_query_kernel, dim3((unsigned)blocks_for(n, 256)), dim3(256), 0, (hipStream_t)stream, child, points,
                     n, offset[0], offset[1], offset[2], invradius[0], invradius[1], invradius[2], packed);
  return check_launch("tree_query");
}

int pxo_tree_relu_sigma(float* data, int64_t n_cells, int data_dim, void* stream) {
  PXO_REQUIRE(n_cells >= 0 && data_dim >= 1, "pxo_tree_relu_sigma: bad sizes");
  if (n_cells == 0) return PXO_OK;
  PXO_REQUIRE(data, "pxo_tree_relu_sigma: null pointer");
  hipLaunchKernelGGL(tree_relu_sigma_kernel, dim3((unsigned)blocks_for(n_cells, 256)), dim3(256), 0, (hipStream_t)stream, data,
                     n_cells, data_dim);
  return check_launch("tree_relu_sigma");
}

static int check_opts(const PxoRenderOpts* o, const char* who) {
  PXO_REQUIRE(o, "%s: null options", who);
  PXO_REQUIRE(o->step_size > 0.0f, "%s: step_size must be > 0 (marching would not advance)", who);
  return PXO_OK;
}

int pxo_grid_weight_workspace_bytes(int reso, size_t* bytes) {
  PXO_REQUIRE(reso >= 1 && reso <= 2048 && bytes, "pxo_grid_weight_workspace_bytes: bad arguments");
  *bytes = (reso % 4 == 0) ? (size_t)2 * reso * reso * reso * sizeof(float) + 256 : 0;   // bricked sigma, bricked weights, a counter
  return PXO_OK;
}

static int g_gw_marcher = -1;   // -1: chosen on the device; 0 / 1: forced (pxo_octree_set_tuning)
static int g_gw_tile_order = 0; // gw_tile: 0 linear, 1 XCD supertiles (PXO_TUNE_GW_TILE_ORDER)

int pxo_grid_weight_render(const float* sigma_grid, int reso, const float* c2w_all, int n_cams, float fx, float fy,
                           int width, int height, const PxoRenderOpts* opts, const float offset[3],
                           const float invradius[3], float* grid_weight, void* ws, size_t ws_bytes, void* stream) {
  if (int rc = check_opts(opts, "pxo_grid_weight_render")) return rc;
  PXO_REQUIRE(reso >= 1 && reso <= 2048 && n_cams >= 0 && width >= 1 && height >= 1, "pxo_grid_weight_render: bad sizes");
  PXO_REQUIRE(fx > 0.0f && fy > 0.0f, "pxo_grid_weight_render: focal length must be > 0");
  if (n_cams == 0) return PXO_OK;
  PXO_REQUIRE(sigma_grid && c2w_all && offset && invradius && grid_weight, "pxo_grid_weight_render: null pointer");
  Vec3 o{{offset[0], offset[1], offset[2]}}, ir{{invradius[0], invradius[1], invradius[2]}};
  const int64_t tiles = (int64_t)((width + 15) / 16) * ((height + 15) / 16);
  PXO_REQUIRE(tiles * n_cams < ((int64_t)1 << 31), "pxo_grid_weight_render: too many tiles for one launch");
  const unsigned gw_grid = (unsigned)(tiles * n_cams);
  hipStream_t s = (hipStream_t)stream;
  const int64_t n = (int64_t)reso * reso * reso;
  if (reso % 4 != 0) {   // grids that do not tile into bricks (never the case for 2^(depth+1), depth >= 1)
    hipLaunchKernelGGL((grid_weight_kernel<false, false>), dim3(gw_grid), dim3(256), 0, s, sigma_grid, reso, c2w_all,
                       n_cams, fx, fy, width, height, *opts, o, ir, reinterpret_cast<int*>(grid_weight));
    return check_launch("grid_weight_render");
  }
  const size_t need = (size_t)2 * n * sizeof(float) + 256;
  if (!ws || ws_bytes < need) {
    set_error("pxo_grid_weight_render: workspace %zu < %zu", ws_bytes, need);
    return PXO_ERR_WORKSPACE;
  }
  float* sigma_b = reinterpret_cast<float*>(ws);
  float* weight_b = sigma_b + n;
  unsigned long long* occupied = reinterpret_cast<unsigned long long*>(weight_b + n);
  if (hipMemsetAsync(weight_b, 0, (size_t)n * sizeof(float) + 256, s) != hipSuccess) {
    set_error("pxo_grid_weight_render: hipMemsetAsync failed");
    return PXO_ERR_HIP;
  }
  const int64_t brick_blocks = blocks_for(n, 256);
  hipLaunchKernelGGL(brick_sigma_kernel, dim3((unsigned)(brick_blocks < 8192 ? brick_blocks : 8192)), dim3(256), 0, s, sigma_grid,
                     reso, sigma_b, opts->sigma_thresh, occupied);
  const bool pow2 = (reso & (reso - 1)) == 0 && reso <= 1024;
  // pxo_octree_set_tuning(PXO_TUNE_GW_MARCHER, 1 | 0) forces the slab-staged / the per-sample marcher (A/B runs, the
  // equality test); default -1: chosen on the device by the fraction of voxels above sigma_thresh (GwSelect).  Slab window width measured at 4 / 5 / 6 bricks:
  // 1.245 / 1.257 / 1.262 ms per camera (the staging is not what bounds the marcher); 6 keeps nearly every sample of a tile inside.
  const int force = g_gw_marcher;
  if (pow2) {
    const int64_t per_cam = gw_blocks_per_cam((width + 15) / 16, (height + 15) / 16, g_gw_tile_order);
    PXO_REQUIRE(per_cam * n_cams < ((int64_t)1 << 31), "pxo_grid_weight_render: too many tiles for one launch");
    hipLaunchKernelGGL(grid_weight_pow2_kernel, dim3((unsigned)(per_cam * n_cams)), dim3(256), 0, s, (const float*)sigma_b, reso, c2w_all, n_cams, fx,
                       fy, width, height, *opts, o, ir, reinterpret_cast<int*>(weight_b), GwSelect{occupied, n, force, g_gw_tile_order});
  }
  else
    hipLaunchKernelGGL((grid_weight_kernel<true, false>), dim3(gw_grid), dim3(256), 0, s, (const float*)sigma_b, reso, c2w_all,
                       n_cams, fx, fy, width, height, *opts, o, ir, reinterpret_cast<int*>(weight_b));
  hipLaunchKernelGGL(unbrick_max_kernel, dim3((unsigned)blocks_for(n, 256)), dim3(256), 0, s, (const float*)weight_b, reso,
                     grid_weight);
  return check_launch("grid_weight_render");
}

// Lanes per ray of the renderer launches.  Measured on the 512^3 / 800x800 / SH16 benchmark (profiles/README.md):
// forward 3.88 / 3.36 / 4.23 ms for 16 / 8 / 4 lanes (8 lanes: twice the rays in flight per wave, half the
// duplicated traversal arithmetic, coefficient rows still 32-byte coalesced); backward 10.9 / 14.0 / 22.1 ms (the
// gradient scatter wants the widest atomic rows); SH25: forward 4.34 / 4.04 ms, backward 13.9 / 17.8 ms for 16 / 8.
// With the forward kernel's 16-byte coefficient loads (round 2: a lane owns groups of 4 consecutive floats) the balance
// moves to fewer lanes: forward 3.51 / 2.98 / 2.44 ms for 16 / 8 / 4 lanes (dword loads: 3.88 / 3.11 / 4.23), SH25 3.45 /
// 3.06 ms for 8 / 4, SH9 2.14 ms at 4; 2 lanes measured slower again (3.25 ms).  At 4 lanes the channel-aligned ownership
// with compile-time K (KF) then takes SH16 to 1.61 ms, SH25 to 2.35 ms, SH9 to 1.75 ms.
// Backward at "4 lanes" is octree_render_bwd4_kernel (4-lane march, 16-lane cooperative scatter): 10.47 ms reusing the
// forward image / 11.20 ms without, against 10.86 / 11.90 ms for the 16-lane kernel (SH25: 14.46 / 15.49 against 14.87 / 16.02).
// So: 4 lanes both ways.  pxo_octree_set_lanes_per_ray forces a value for A/B runs and the per-instantiation tests.
static int g_row_override[2] = {0, 0};   // [forward, backward]; 0 = the measured default
static int render_row(bool backward, int data_dim) {
  if (g_row_override[backward ? 1 : 0]) return g_row_override[backward ? 1 : 0];
  (void)data_dim;
  return 4;
}

// Rows of the backward kernel's per-wave write-combining cache.  Measured (800x800, SH16, reusing the forward image):
// 9.7 ms direct scatter, 5.0 / 4.8 / 4.4 / 4.3 ms with 4 / 8 / 16 / 32 rows; two-march form 10.3 -> 6.3 / 6.1 / 5.6 / 6.3 (32 rows
// cost occupancy: 35 KB of LDS per workgroup).  pxo_octree_set_tuning(PXO_TUNE_BWD_CACHE_ROWS, 0 | 4 | 8 | 16 | 32 | 64)
// selects another instantiation for A/B runs (0 = direct scatter); anything else is rejected there.
static int g_bwd_wc_rows = 16;
static int g_bwd_update = 0;     // PXO_TUNE_BWD_UPDATE: 0 the whole wave on one sample's row, 1 every ray's 4 lanes on its own row
static int bwd_wc_slots() { return g_bwd_wc_rows; }

static int render_args(const PxoTree* tree, const PxoCamera* cam, const float* origins, const float* dirs,
                       const float* viewdirs, int64_t B, const PxoRenderOpts* opts, const char* who, bool backward,
                       RenderArgs& A, unsigned& grid, int& row) {
  if (int rc = check_opts(opts, who)) return rc;
  PXO_REQUIRE(tree && tree->child && tree->data, "%s: null tree", who);
  const int K = tree->basis_dim;
  PXO_REQUIRE(K == 1 || K == 4 || K == 9 || K == 16 || K == 25, "%s: basis_dim %d is not an SH format (1,4,9,16,25)", who, K);
  PXO_REQUIRE(tree->data_dim == 3 * K + 1, "%s: data_dim %d != 3*basis_dim+1", who, tree->data_dim);
  PXO_REQUIRE(tree->n_internal >= 1 && tree->n_internal < ((int64_t)1 << 28), "%s: n_internal out of range", who);
  PXO_REQUIRE(B >= 0, "%s: B < 0", who);
  row = render_row(backward, tree->data_dim);
  A.tree = *tree;
  A.opt = *opts;
  A.B = B;
  A.has_cam = cam != nullptr;
  int64_t blocks;
  if (cam) {
    PXO_REQUIRE(cam->c2w && cam->width >= 1 && cam->height >= 1 && cam->fx > 0.0f && cam->fy > 0.0f, "%s: bad camera", who);
    PXO_REQUIRE(B == (int64_t)cam->width * cam->height, "%s: B must be width*height in camera mode", who);
    A.cam = *cam;
    A.origins = A.dirs = A.viewdirs = nullptr;
    const int rpw = 64 / row, wtx = rpw <= 4 ? 2 : (rpw <= 16 ? 4 : 8), tx = 2 * wtx, ty = 2 * (rpw / wtx);
    blocks = (int64_t)((cam->width + tx - 1) / tx) * ((cam->height + ty - 1) / ty);
  } else {
    PXO_REQUIRE(B == 0 || (origins && dirs && viewdirs), "%s: null ray arrays", who);
    A.cam = PxoCamera{};
    A.origins = origins; A.dirs = dirs; A.viewdirs = viewdirs;
    blocks = blocks_for(B, kRenderThreads / row);
  }
  PXO_REQUIRE(blocks < ((int64_t)1 << 31), "%s: too many rays for one launch", who);
  grid = (unsigned)blocks;
  return PXO_OK;
}

int pxo_octree_set_lanes_per_ray(int forward, int backward) {
  auto ok = [](int v) { return v == 0 || v == 4 || v == 8 || v == 16; };
  PXO_REQUIRE(ok(forward) && ok(backward), "pxo_octree_set_lanes_per_ray: lanes must be 0 (default), 4, 8 or 16");
  g_row_override[0] = forward;
  g_row_override[1] = backward;
  return PXO_OK;
}

int pxo_octree_set_tuning(int knob, int value) {
  switch (knob) {
    case PXO_TUNE_GW_MARCHER:
      PXO_REQUIRE(value >= -1 && value <= 1, "pxo_octree_set_tuning: marcher must be -1 (chosen on the device), 0 (per-sample) or 1 (slab-staged)");
      g_gw_marcher = value;
      return PXO_OK;
    case PXO_TUNE_BWD_CACHE_ROWS:
      PXO_REQUIRE(value == 0 || value == 4 || value == 8 || value == 16 || value == 32 || value == 64,
                  "pxo_octree_set_tuning: write-combining rows must be 0, 4, 8, 16, 32 or 64 (got %d)", value);
      g_bwd_wc_rows = value;
      return PXO_OK;
    case PXO_TUNE_GW_TILE_ORDER:
      PXO_REQUIRE(value == 0 || value == 1, "pxo_octree_set_tuning: weight-mask tile order must be 0 (linear) or 1 (XCD supertiles), got %d", value);
      g_gw_tile_order = value;
      return PXO_OK;
    case PXO_TUNE_BWD_UPDATE:
      PXO_REQUIRE(value == 0 || value == 1, "pxo_octree_set_tuning: backward update must be 0 (wave per sample) or 1 (ray-parallel), got %d", value);
      g_bwd_update = value;
      return PXO_OK;
    default:
      set_error("pxo_octree_set_tuning: unknown knob %d", knob);
      return PXO_ERR_ARG;
  }
}

int pxo_octree_get_tuning(int knob, int* value) {
  PXO_REQUIRE(value != nullptr, "pxo_octree_get_tuning: NULL pointer");
  switch (knob) {
    case PXO_TUNE_GW_MARCHER: *value = g_gw_marcher; return PXO_OK;
    case PXO_TUNE_BWD_CACHE_ROWS: *value = g_bwd_wc_rows; return PXO_OK;
    case PXO_TUNE_BWD_UPDATE: *value = g_bwd_update; return PXO_OK;
    case PXO_TUNE_GW_TILE_ORDER: *value = g_gw_tile_order; return PXO_OK;
    default: set_error("pxo_octree_get_tuning: unknown knob %d", knob); return PXO_ERR_ARG;
  }
}

int pxo_octree_render_fwd(const PxoTree* tree, const PxoCamera* cam, const float* origins, const float* dirs,
                          const float* viewdirs, int64_t B, const PxoRenderOpts* opts, float* out_rgb, void* stream) {
  RenderArgs A;
  unsigned grid;
  int row;
  if (int rc = render_args(tree, cam, origins, dirs, viewdirs, B, opts, "pxo_octree_render_fwd", false, A, grid, row)) return rc;
  if (B == 0) return PXO_OK;
  PXO_REQUIRE(out_rgb, "pxo_octree_render_fwd: null output");
  const float* none = nullptr;
  switch (row) {
    case 4:
#define PXO_FWD4(KF_) hipLaunchKernelGGL((octree_render_kernel<0, 4, true, KF_>), dim3(grid), dim3(kRenderThreads), 0, (hipStream_t)stream, A, out_rgb, none, none, (float*)nullptr)
      switch (tree->basis_dim) {
        case 1: PXO_FWD4(1); break;
        case 4: PXO_FWD4(4); break;
        case 9: PXO_FWD4(9); break;
        case 16: PXO_FWD4(16); break;
        case 25: PXO_FWD4(25); break;
        default: PXO_FWD4(-1); break;
      }
#undef PXO_FWD4
      break;
    case 8: hipLaunchKernelGGL((octree_render_kernel<0, 8, true>), dim3(grid), dim3(kRenderThreads), 0, (hipStream_t)stream, A, out_rgb, none, none, (float*)nullptr); break;
    default: hipLaunchKernelGGL((octree_render_kernel<0, 16, true>), dim3(grid), dim3(kRenderThreads), 0, (hipStream_t)stream, A, out_rgb, none, none, (float*)nullptr); break;
  }
  return check_launch("octree_render_fwd");
}

int pxo_octree_render_bwd(const PxoTree* tree, const PxoCamera* cam, const float* origins, const float* dirs,
                          const float* viewdirs, int64_t B, const PxoRenderOpts* opts, const float* out_rgb,
                          const float* grad_out, float* grad_data, void* stream) {
  RenderArgs A;
  unsigned grid;
  int row;
  if (int rc = render_args(tree, cam, origins, dirs, viewdirs, B, opts, "pxo_octree_render_bwd", true, A, grid, row)) return rc;
  if (B == 0) return PXO_OK;
  PXO_REQUIRE(grad_out && grad_data, "pxo_octree_render_bwd: null pointer");
  // the gradient pass marches exactly (no early stop, no 1/(1-T) rescale): a forward image made with
  // stop_thresh > 0 is not the image whose gradient this is
  PXO_REQUIRE(out_rgb == nullptr || opts->stop_thresh == 0.0f,
              "pxo_octree_render_bwd: out_rgb must come from an exact march (stop_thresh == 0), got stop_thresh %g",
              (double)opts->stop_thresh);
  switch (row) {
    case 4:
#define PXO_BWD4W(KF_, WC_)                                                                                                         \
  do {                                                                                                                            \
    if (g_bwd_update == 1 && WC_ > 0)                                                                                             \
      hipLaunchKernelGGL((octree_render_bwd4_kernel<KF_, WC_, (WC_ > 0 ? 1 : 0)>), dim3(grid), dim3(kRenderThreads), 0,             \
                         (hipStream_t)stream, A, out_rgb, grad_out, grad_data);                                                   \
    else                                                                                                                          \
      hipLaunchKernelGGL((octree_render_bwd4_kernel<KF_, WC_, 0>), dim3(grid), dim3(kRenderThreads), 0, (hipStream_t)stream, A,    \
                         out_rgb, grad_out, grad_data);                                                                          \
  } while (0)
#define PXO_BWD4(KF_)                                     \
  switch (bwd_wc_slots()) {                               \
    case 4: PXO_BWD4W(KF_, 4); break;                     \
    case 8: PXO_BWD4W(KF_, 8); break;                     \
    case 16: PXO_BWD4W(KF_, 16); break;                   \
    case 32: PXO_BWD4W(KF_, 32); break;                   \
    case 64: PXO_BWD4W(KF_, 64); break;                   \
    default: PXO_BWD4W(KF_, 0); break;                    \
  }
      switch (tree->basis_dim) {
        case 16: PXO_BWD4(16); break;
        case 25: PXO_BWD4(25); break;
        default: PXO_BWD4(-1); break;
      }
#undef PXO_BWD4W
#undef PXO_BWD4
      break;
    case 8: hipLaunchKernelGGL((octree_render_kernel<1, 8>), dim3(grid), dim3(kRenderThreads), 0, (hipStream_t)stream, A, (float*)nullptr, out_rgb, grad_out, grad_data); break;
    default: hipLaunchKernelGGL((octree_render_kernel<1, 16>), dim3(grid), dim3(kRenderThreads), 0, (hipStream_t)stream, A, (float*)nullptr, out_rgb, grad_out, grad_data); break;
  }
  return check_launch("octree_render_bwd");
}

int pxo_octree_count_work(const PxoTree* tree, const PxoCamera* cam, const PxoRenderOpts* opts, unsigned long long* counts,
                          uint8_t* leaf_seen, void* stream) {
  PXO_REQUIRE(cam != nullptr && counts != nullptr, "pxo_octree_count_work: needs a camera and a counts[4] buffer");
  RenderArgs A;
  unsigned grid;
  int row;
  if (int rc = render_args(tree, cam, nullptr, nullptr, nullptr, (int64_t)cam->width * cam->height, opts,
                           "pxo_octree_count_work", false, A, grid, row)) return rc;
  const int64_t rays = (int64_t)cam->width * cam->height;
  hipLaunchKernelGGL(octree_count_kernel, dim3((unsigned)blocks_for(rays, 256)), dim3(256), 0, (hipStream_t)stream, A, counts,
                     leaf_seen);
  return check_launch("octree_count_work");
}

int pxo_grid_weight_count_work(const float* sigma_grid, int reso, const float* c2w_all, int n_cams, float fx, float fy,
                               int width, int height, const PxoRenderOpts* opts, const float offset[3],
                               const float invradius[3], unsigned long long* counts, uint8_t* voxel_seen, void* stream) {
  if (int rc = check_opts(opts, "pxo_grid_weight_count_work")) return rc;
  PXO_REQUIRE(sigma_grid && c2w_all && offset && invradius && counts && reso >= 1 && n_cams >= 1 && width >= 1 && height >= 1,
              "pxo_grid_weight_count_work: bad arguments");
  // the counting march divides by the grid size as the power-of-two weight-mask kernels do (an exact multiplication); for
  // other sizes the renderer's own step arithmetic differs in the last bit and the counted sample sequence could too
  PXO_REQUIRE((reso & (reso - 1)) == 0 && reso >= 4, "pxo_grid_weight_count_work: reso %d is not a power of two >= 4 "
              "(the counters repeat the march of the power-of-two weight-mask kernels only)", reso);
  Vec3 o, ir;
  for (int a = 0; a < 3; ++a) { o.v[a] = offset[a]; ir.v[a] = invradius[a]; }
  const int64_t rays = (int64_t)width * height * n_cams;
  hipLaunchKernelGGL(grid_count_kernel, dim3((unsigned)blocks_for(rays, 256)), dim3(256), 0, (hipStream_t)stream, sigma_grid, reso,
                     c2w_all, n_cams, fx, fy, width, height, *opts, o, ir, counts, voxel_seen);
  return check_launch("grid_weight_count_work");
}

int pxo_image_mse(const float* im, const float* gt, int64_t n, float* grad, float* sse_out, void* stream) {
  PXO_REQUIRE(n >= 1, "pxo_image_mse: n < 1");
  PXO_REQUIRE(im && gt && sse_out, "pxo_image_mse: null pointer");
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(sse_out, 0, sizeof(float), s) != hipSuccess) {
    set_error("pxo_image_mse: hipMemsetAsync failed");
    return PXO_ERR_HIP;
  }
  const int64_t blocks = blocks_for(n, 256 * 8);
  hipLaunchKernelGGL(image_mse_kernel, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, s, im, gt, n, grad, sse_out);
  return check_launch("image_mse");
}

int pxo_sgd_step(float* params, const float* grads, float* momentum_buf, int64_t n, float lr, float mu, int nesterov,
                 int first_step, void* stream) {
  PXO_REQUIRE(n >= 0, "pxo_sgd_step: n < 0");
  if (n == 0) return PXO_OK;
  PXO_REQUIRE(params && grads, "pxo_sgd_step: null pointer");
  PXO_REQUIRE(mu == 0.0f || momentum_buf, "pxo_sgd_step: momentum needs a buffer");
  hipLaunchKernelGGL(sgd_kernel, dim3((unsigned)blocks_for(n, 256)), dim3(256), 0, (hipStream_t)stream, params, grads,
                     momentum_buf, n, lr, mu, nesterov, first_step);
  return check_launch("sgd_step");
}

}  // extern "C"
