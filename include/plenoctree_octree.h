/*
 * plenoctree_octree.h -- C ABI of the PlenOctree side of the path on MI355X (gfx950): building the
 * N3Tree from the dense-grid evaluation, the training-view weight mask, and the octree volume
 * renderer with its gradient.  Same library (libplenoctree_hip.so) and conventions as
 * plenoctree_hip.h: extern "C", plain pointers, caller-owned device memory, asynchronous on
 * `void* stream` unless a host-side result is returned, 0 = ok / negative = error.
 *
 * In the reference these operations are calls into the third-party package svox
 * (svox>=0.2.28, not part of the reference tree); each entry point cites the reference call
 * site (file:line) whose svox call it replaces.
 *
 * N3Tree storage (N = 2), identical to svox's and to the npz keys of octree/compression.py:76-86:
 *   child        int32 [n, 2,2,2]        index(child node) - n, 0 = leaf
 *   parent_depth int32 [n, 2]            (packed parent cell index ((p*2+i)*2+j)*2+k, depth)
 *   data         float [n, 2,2,2, D]     D = 3*basis_dim + 1, rgb SH coefficients channel-major,
 *                                        last channel sigma
 *   offset[3], invradius[3]              x_tree = offset + invradius * x_world
 * Nodes are stored breadth-first: all nodes of depth d before depth d+1, each level ordered by
 * packed parent cell index (the order svox's refine() produces when whole levels are refined).
 */
#ifndef PLENOCTREE_OCTREE_H_
#define PLENOCTREE_OCTREE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PXO_TREE_MAX_DEPTH 10   /* depth_limit <= 10: grid of at most 2048^3 */

/* svox RenderOptions (VolumeRenderer._get_options; octree/extraction.py:184-195). */
typedef struct PxoRenderOpts {
  float step_size;              /* --renderer_step_size, octree/nerf/utils.py:211-215 */
  float background_brightness;  /* 1.0 */
  float sigma_thresh;           /* 0 exact / 1e-2 fast (eval_octree: fast = not no_early_stop) */
  float stop_thresh;            /* 0 exact / 1e-2 fast */
} PxoRenderOpts;

/* Read-only view of a tree for the renderers. */
typedef struct PxoTree {
  const int32_t* child;
  const float* data;
  int64_t n_internal;
  int32_t data_dim;             /* 3*basis_dim + 1 */
  int32_t basis_dim;            /* 1, 4, 9, 16 or 25 (data_format SH<k>) */
  float offset[3];
  float invradius[3];
} PxoTree;

/* Pinhole camera (svox CameraSpec; octree/extraction.py:197-201, octree/nerf/utils.py:473-474).
 * c2w: device pointer to the first 3 rows of the camera-to-world matrix, 12 floats row-major. */
typedef struct PxoCamera {
  const float* c2w;
  float fx, fy;
  int32_t width, height;
} PxoCamera;

/* ---- step 1: mask -> tree  (octree/extraction.py:322-352: mask, `tree[grid].refine()` x depth) ---- */

/* mask[i] = value[i] >= thresh (sigma or grid-weight masking, octree/extraction.py:322-331). */
int pxo_threshold_mask(const float* value, int64_t n, float thresh, uint8_t* mask, void* stream);

/* Workspace for the occupancy pyramid of a 2^(depth+1) grid. */
int pxo_tree_workspace_bytes(int depth, size_t* bytes);
/* Builds the occupancy pyramid of `mask` ([reso^3] bytes, reso = 2^(depth+1), x slowest) in `ws` and
 * returns on the HOST the number of nodes per depth, level_nodes[0..depth] (level_nodes[0] = 1, the
 * root).  Synchronises `stream`.  The tree will have sum(level_nodes) nodes. */
int pxo_tree_count_nodes(const uint8_t* mask, int depth, void* ws, size_t ws_bytes, int64_t* level_nodes,
                         void* stream);
/* Writes child [n,8] and parent_depth [n,2] of the tree whose pyramid is in `ws` (after
 * pxo_tree_count_nodes on the same ws); n = sum(level_nodes). */
int pxo_tree_build(const void* ws, size_t ws_bytes, int depth, const int64_t* level_nodes, int32_t* child,
                   int32_t* parent_depth, void* stream);

/* ---- step 2: leaf samples and assignment  (octree/extraction.py:355-394, :503) ---- */

/* tree[inds].sample(S) for the 8 cells of each node in [node0, node0+n_nodes): point (cell c, sample s)
 * = corner + u * side in tree coordinates, returned in world coordinates ((p - offset) / invradius).
 * u: [n_nodes*8*S, 3] uniforms in [0,1) (pxo_uniform).  points: [n_nodes*8*S, 3], cells in packed order. */
int pxo_tree_sample_cells(const int32_t* parent_depth, int64_t node0, int64_t n_nodes, int S, const float* u,
                          const float offset[3], const float invradius[3], float* points, void* stream);
/* The same for arbitrary leaves given by packed cell index node*8 + cell (`tree[leaf_inds].sample(S)` with any index
 * set, octree/extraction.py:358-369): u [n_cells*S, 3], points [n_cells*S, 3]. */
int pxo_tree_sample_leaves(const int32_t* parent_depth, const int64_t* packed, int64_t n_cells, int S, const float* u,
                           const float offset[3], const float invradius[3], float* points, void* stream);
/* tree[points] (the indexing behind `tree[grid].refine()`, octree/extraction.py:341-350): packed index
 * node*8 + cell of the leaf that contains each world-space point [n,3] (tree coordinates clamped to
 * [0, 1 - 1e-6] as svox does). */
int pxo_tree_query(const int32_t* child, const float* points, int64_t n, const float offset[3],
                   const float invradius[3], int64_t* packed, void* stream);
/* tree[:, -1:].relu_()  (octree/extraction.py:503): clamps the sigma channel of n_cells cells at 0. */
int pxo_tree_relu_sigma(float* data, int64_t n_cells, int data_dim, void* stream);

/* ---- weight mask  (_C.grid_weight_render, octree/extraction.py:181-214) ---- */

/* For each of the n_cams cameras (c2w_all: [n_cams,12] device) and every pixel, marches the ray through
 * the dense sigma grid [reso^3] and keeps, per voxel, the maximum compositing weight
 * light * (1 - exp(-dt * sigma)).  grid_weight [reso^3] must be zero-initialised by the caller for the
 * first call; successive calls accumulate the maximum (torch.max over cameras, :206-212).
 * ws: pxo_grid_weight_workspace_bytes(reso) bytes (brick-ordered copies of the grid and the weights, and the count of
 * voxels above sigma_thresh by which the launch picks its marcher on the device: slab-staged for dense grids, per-sample
 * for sparse ones - same samples, same arithmetic, bit-identical weights either way; the call stays asynchronous). */
int pxo_grid_weight_workspace_bytes(int reso, size_t* bytes);
int pxo_grid_weight_render(const float* sigma_grid, int reso, const float* c2w_all, int n_cams, float fx,
                           float fy, int width, int height, const PxoRenderOpts* opts, const float offset[3],
                           const float invradius[3], float* grid_weight, void* ws, size_t ws_bytes, void* stream);

/* ---- octree volume renderer  (VolumeRenderer.render_persp / render; octree/nerf/utils.py:456-474,
 *      octree/optimization.py:174-216) ---- */

/* Lanes of a wave that cooperate on one ray in the forward / backward renderer launches: 4, 8 or 16, or 0 for
 * the measured default (4 both ways; backward at 4 = 4-lane march with a 16-lane cooperative gradient scatter).  A tuning knob (results are identical up to the SH summation
 * order); process-wide. */
int pxo_octree_set_lanes_per_ray(int forward, int backward);

/* The other two launch-time choices between kernels that compute the SAME result (A/B sessions, the bit-equality test of
 * the two weight-mask marchers); process-wide, validated, no environment variables are read anywhere in the library.
 *   PXO_TUNE_GW_MARCHER      -1 chosen on the device by the occupied fraction of the grid (default), 0 per-sample, 1 slab-staged
 *   PXO_TUNE_BWD_CACHE_ROWS  rows of the backward renderer's per-wave write-combining cache: 16 (default), 0 (direct
 *                            scatter), 4, 8, 32, 64
 *   PXO_TUNE_BWD_UPDATE      how a sample's gradient row reaches that cache: 0 the whole wave on one sample's row at a time,
 *                            1 every ray's own 4 lanes on its row, all 16 rays of the wave at once (LDS float atomics; misses
 *                            elect one winner per slot).  Same sums in a different order (float32 round-off apart). */
#define PXO_TUNE_GW_MARCHER 0
#define PXO_TUNE_BWD_CACHE_ROWS 1
#define PXO_TUNE_BWD_UPDATE 2
#define PXO_TUNE_GW_TILE_ORDER 3   /* weight mask on power-of-two grids: 0 workgroups take tiles in row-major order (default), 1 the tiles of
                                      a 4x4-tile square all go to one XCD (one L2); same weights either way (max is order-free) */
int pxo_octree_set_tuning(int knob, int value);
int pxo_octree_get_tuning(int knob, int* value);

/* Forward.  Rays come either from `cam` (cam != NULL: B must be width*height, ray r = pixel
 * (r % width, r / width), out [H,W,3]) or from explicit arrays origins/dirs/viewdirs [B,3] in world
 * space with unit dirs (cam == NULL).  out_rgb [B,3]. */
int pxo_octree_render_fwd(const PxoTree* tree, const PxoCamera* cam, const float* origins, const float* dirs,
                          const float* viewdirs, int64_t B, const PxoRenderOpts* opts, float* out_rgb,
                          void* stream);
/* Gradient of sum(out_rgb * grad_out) w.r.t. tree->data, ACCUMULATED (atomic adds) into grad_data
 * [n_internal,2,2,2,D] -- zero it first (optimizer.zero_grad(), octree/optimization.py:221-224).
 * Training semantics: exact marching (stop_thresh is ignored: no early-stop rescale).
 * out_rgb: the [B,3] result of pxo_octree_render_fwd for the same rays with the same exact options, or NULL
 * (the kernel then re-marches once more to recover it).  Passing out_rgb together with opts->stop_thresh > 0
 * is rejected (PXO_ERR_ARG): an early-stopped image is not the image this gradient belongs to. */
int pxo_octree_render_bwd(const PxoTree* tree, const PxoCamera* cam, const float* origins, const float* dirs,
                          const float* viewdirs, int64_t B, const PxoRenderOpts* opts, const float* out_rgb,
                          const float* grad_out, float* grad_data, void* stream);


/* ---- work counters of the marchers (roofline pass: scripts/octree_bench.py states each kernel's algorithmic bytes from
 *      these; nothing on the product path calls them) ----
 * One thread per ray repeats the renderer's / the weight mask's march -- same ray set-up, leaf lookup, step rule and early
 * stop, hence the same sample sequence -- and counts:
 *   counts[0] rays that enter the volume        counts[2] samples with sigma > sigma_thresh (one leaf row read / one
 *   counts[1] samples (one sigma read each)               weight update each)
 *   counts[3] child-pointer loads of the leaf lookups (tree marcher only)
 * counts: 4 device uint64, ACCUMULATED (zero them first).  leaf_seen [n_internal*8] / voxel_seen [reso^3] (uint8, may be
 * NULL): set to 1 where a sample above the threshold fell, so that the distinct rows a launch must fetch from HBM at
 * least once can be counted.  pxo_grid_weight_count_work takes power-of-two grids only (reso >= 4: every grid the extraction
 * makes): it repeats the march of the power-of-two weight-mask kernels, whose division by reso is an exact multiplication. */
int pxo_octree_count_work(const PxoTree* tree, const PxoCamera* cam, const PxoRenderOpts* opts,
                          unsigned long long* counts, uint8_t* leaf_seen, void* stream);
int pxo_grid_weight_count_work(const float* sigma_grid, int reso, const float* c2w_all, int n_cams, float fx, float fy,
                               int width, int height, const PxoRenderOpts* opts, const float offset[3],
                               const float invradius[3], unsigned long long* counts, uint8_t* voxel_seen,
                               void* stream);

/* mse = mean((clamp(im,0,1) - gt)^2) and its gradient w.r.t. im  (octree/optimization.py:217-219);
 * n = number of floats.  sse_out: device scalar receiving sum of squares (mse = sse/n); grad may be NULL. */
int pxo_image_mse(const float* im, const float* gt, int64_t n, float* grad, float* sse_out, void* stream);

/* torch.optim.SGD step on the tree data (octree/optimization.py:176-181): with momentum mu > 0,
 * buf = mu*buf + g (buf = g on the first step), g' = g + mu*buf if nesterov else buf; p -= lr * g'. */
int pxo_sgd_step(float* params, const float* grads, float* momentum_buf, int64_t n, float lr, float mu,
                 int nesterov, int first_step, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PLENOCTREE_OCTREE_H_ */
