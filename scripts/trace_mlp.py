#!/usr/bin/env python
"""Cycle stamps of the fused MLP forward kernel, from an instrumented COPY of the kernel source (the shipped sources carry no
trace hooks): `python scripts/trace_mlp.py build` patches plenoctree_amd/csrc/mlp_kernels.hip into scripts/trace_build/,
and compiles plenoctree_amd/libplenoctree_hip_trace.so; `python scripts/trace_mlp.py run` (on the GPU box, with
PXO_ALLOW_VARIANT=1 PXO_LIB=.../libplenoctree_hip_trace.so) runs the training forward on 4096 x 192 + 10000 rows and prints the
average cycles per phase of one wave of workgroup 0 over its steady-state tiles.

Stamps (s_memtime, wave 0 and wave 4 of workgroup 0, kept in LDS, flushed at kernel end): per layer l
  0 before the GEMM   1 after the GEMM   2 after the post-GEMM barrier   3 after the epilogue's LDS writes + mask stores
  4 after the tile-column stores are issued   5 after the post-epilogue barrier
and 14 / 15 at the start of a tile / after the encoding barrier.
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "plenoctree_amd", "csrc")
OUT = os.path.join(ROOT, "scripts", "trace_build")
LIB = os.path.join(ROOT, "plenoctree_amd", "libplenoctree_hip_trace.so")

PREAMBLE = r'''
// ---- injected by scripts/trace_mlp.py -------------------------------------------------------
__device__ unsigned long long g_trace[2][4096];
__device__ int g_trace_n[2];
#define STAMP(k)                                                                                          \
  do {                                                                                                    \
    if (trace_on && s_tn < 1536) {                                                                        \
      s_trace[s_tn] = ((unsigned long long)(((unsigned)(l) << 4) | (unsigned)(k)) << 48) |               \
                      (__builtin_readcyclecounter() & 0xFFFFFFFFFFFFull);                                 \
      ++s_tn;                                                                                             \
    }                                                                                                     \
  } while (0)
'''


def patch(src):
    s = src
    s = s.replace("namespace pxo {\n", "namespace pxo {\n" + PREAMBLE, 1)
    # the mid-loop stamp of the skew experiment
    s = s.replace("    if (mid_barrier && g + 4 == kgroups / 2) { lds_barrier(); PXO_PIN(); }", "    if (mid_barrier && g + 4 == kgroups / 2) { lds_barrier(); PXO_PIN(); }", 1)
    # forward tile: trace state is handed in through two extra parameters
    s = s.replace("float* __restrict__ enc_out, uint32_t* __restrict__ mask, int tid, int lane,\n                                         int wave) {",
                  "float* __restrict__ enc_out, uint32_t* __restrict__ mask, int tid, int lane,\n                                         int wave, unsigned long long* s_trace, int& s_tn, bool trace_on) {", 1)
    s = s.replace("  lds_barrier();   // previous tile's head GEMM has consumed the LDS tile\n  float enc_keep[EncGeom<RBN>::kColsPer];\n  posenc_tile<RBN, !SAVE, !SAVE>(lds, pts, grid, row0, M, tid, &enc_keep);\n  lds_barrier();",
                  "  { const int l = 15; STAMP(14); }\n  lds_barrier();   // previous tile's head GEMM has consumed the LDS tile\n  float enc_keep[EncGeom<RBN>::kColsPer];\n  posenc_tile<RBN, !SAVE, !SAVE>(lds, pts, grid, row0, M, tid, &enc_keep);\n  lds_barrier();\n  { const int l = 15; STAMP(15); }", 1)
    s = s.replace("    if (SAVE && l > 0) {\n      // the previous layer's activations leave", "    STAMP(0);\n    if (SAVE && l > 0) {\n      // the previous layer's activations leave", 1)
    s = s.replace("    if (l == 5) {\n      // skip connection", "    if (l != 5) STAMP(1);\n    if (l == 5) {\n      // skip connection", 1)
    s = s.replace("      gemm_lds_packed<RBN, kCB>(arow, wimg, wp + 32 * 8 * 64, 8, 8 * 64, acc, bfrag);\n    }",
                  "      gemm_lds_packed<RBN, kCB>(arow, wimg, wp + 32 * 8 * 64, 8, 8 * 64, acc, bfrag);\n      STAMP(1);\n    }", 1)
    s = s.replace("    lds_barrier();  // every wave has consumed the columns this wave is about to rewrite\n    // re-derive the lane ids",
                  "    lds_barrier();  // every wave has consumed the columns this wave is about to rewrite\n    STAMP(2);\n    // re-derive the lane ids", 1)
    s = s.replace("      for (int w = 0; w < kWordsUsed; ++w) mp[w] = mw[w];\n      // layers 0..6 leave for HBM",
                  "      for (int w = 0; w < kWordsUsed; ++w) mp[w] = mw[w];\n      STAMP(3);\n      // layers 0..6 leave for HBM", 1)
    s = s.replace("      if (l == kDepth - 1) store_wave_cols<RBN>(lds, acts + (int64_t)l * M * kW, row0, M, wave, lane_e);\n    }\n    lds_barrier();\n  }",
                  "      if (l == kDepth - 1) store_wave_cols<RBN>(lds, acts + (int64_t)l * M * kW, row0, M, wave, lane_e);\n      STAMP(4);\n    }\n    lds_barrier();\n    STAMP(5);\n  }", 1)
    # kernel: trace buffers + flush
    s = s.replace("  __shared__ __attribute__((aligned(16))) float lds[kTM * kLDA];\n  const int tid = threadIdx.x, lane = tid & 63;\n  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);\n  for (int64_t tile = blockIdx.x; tile < ts.n_full; tile += gridDim.x)\n    fwd_tile<NHB, SAVE, RGB, kRB>(lds, pk, pts, grid, M, deg, tile * kTM, tile, raw_rgb, raw_sigma, acts, enc_out,\n                                  mask, tid, lane, wave);",
                  "  __shared__ __attribute__((aligned(16))) float lds[kTM * kLDA];\n  __shared__ unsigned long long s_trace_all[2][1536];\n  const int tid = threadIdx.x, lane = tid & 63;\n  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);\n  const bool trace_on = blockIdx.x == 0 && lane == 0 && (wave == 0 || wave == 4);\n  unsigned long long* s_trace = s_trace_all[wave >> 2];\n  int s_tn = 0;\n  for (int64_t tile = blockIdx.x; tile < ts.n_full; tile += gridDim.x) {\n    fwd_tile<NHB, SAVE, RGB, kRB>(lds, pk, pts, grid, M, deg, tile * kTM, tile, raw_rgb, raw_sigma, acts, enc_out,\n                                  mask, tid, lane, wave, s_trace, s_tn, trace_on);\n  }", 1)
    s = s.replace("    fwd_tile<NHB, SAVE, RGB, kRB / 2>(lds, pk, pts, grid, M, deg, ts.half_row0 + h * (kTM / 2), ts.n_full + h,\n                                      raw_rgb, raw_sigma, acts, enc_out, mask, tid, lane, wave);\n}",
                  "    fwd_tile<NHB, SAVE, RGB, kRB / 2>(lds, pk, pts, grid, M, deg, ts.half_row0 + h * (kTM / 2), ts.n_full + h,\n                                      raw_rgb, raw_sigma, acts, enc_out, mask, tid, lane, wave, s_trace, s_tn, false);\n  if (trace_on) {\n    const int w = wave >> 2;\n    for (int i = 0; i < s_tn; ++i) g_trace[w][i] = s_trace[i];\n    g_trace_n[w] = s_tn;\n  }\n}", 1)
    s = s.replace("}  // namespace pxo", '''}  // namespace pxo
extern "C" int pxo_debug_trace(unsigned long long* out, int which, int cap) {
  int n = 0;
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(&n, HIP_SYMBOL(pxo::g_trace_n), sizeof(int), which * sizeof(int));
  if (n > cap) n = cap;
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(pxo::g_trace), sizeof(unsigned long long) * n, (size_t)which * 4096 * sizeof(unsigned long long));
  return n;
}''', 1)
    return s


def patch_skew(s):
    """EXPERIMENT (round 3, not shipped): waves 0-3 / 4-7 run the trunk layers one barrier-delimited phase apart -- GEMM over
    columns 0-127, GEMM over columns 128-255, epilogue -- so that a group's epilogue runs beside the other group's MFMAs."""
    s = s.replace("                                                f32x4 (&b)[4][CBN], const TileCopy* tc = nullptr) {\n  constexpr int D = kBDist;",
                  "                                                f32x4 (&b)[4][CBN], const TileCopy* tc = nullptr, bool mid_barrier = false) {\n  constexpr int D = kBDist;", 1)
    s = s.replace("    load_a<RBN>(arow, cl(g + 4), a0);\n    load_b<CBN>(w, wu, cl(g + 3 + D), kg_stride, b[(3 + D) & 3]);",
                  "    if (mid_barrier && g + 4 == kgroups / 2) { lds_barrier(); PXO_PIN(); }\n    load_a<RBN>(arow, cl(g + 4), a0);\n    load_b<CBN>(w, wu, cl(g + 3 + D), kg_stride, b[(3 + D) & 3]);", 1)
    s = s.replace("  for (int l = 0; l < kDepth; ++l) {\n    // the accumulators start from the bias",
                  "  const int grp = wave >> 2;\n  for (int l = 0; l < kDepth; ++l) {\n    const bool skewed = l != 0 && l != 5;\n    if (grp == 1 && (l == 1 || l == 6)) lds_barrier();\n    // the accumulators start from the bias", 1)
    s = s.replace("    gemm_lds_packed<RBN, kCB>(arow, wimg, wp, l == 0 ? 8 : 32, 8 * 64, acc, bfrag);",
                  "    gemm_lds_packed<RBN, kCB>(arow, wimg, wp, l == 0 ? 8 : 32, 8 * 64, acc, bfrag, nullptr, skewed);", 1)
    s = s.replace("      STAMP(4);\n    }\n    lds_barrier();\n    STAMP(5);\n  }",
                  "      STAMP(4);\n    }\n    lds_barrier();\n    STAMP(5);\n    if (grp == 0 && (l == 4 || l == 7)) { lds_barrier(); STAMP(6); }\n  }", 1)
    s = s.replace("    if (grp == 1 && (l == 1 || l == 6)) lds_barrier();", "    if (grp == 1 && (l == 1 || l == 6)) { lds_barrier(); STAMP(7); }", 1)
    if "--prio" in sys.argv:       # the epilogue wave outranks its SIMD partner's MFMA stream
        s = s.replace("    STAMP(2);\n    // re-derive the lane ids", "    STAMP(2);\n    __builtin_amdgcn_s_setprio(3);\n    // re-derive the lane ids", 1)
        s = s.replace("      STAMP(4);\n    }\n    lds_barrier();\n    STAMP(5);", "      STAMP(4);\n    }\n    __builtin_amdgcn_s_setprio(0);\n    lds_barrier();\n    STAMP(5);", 1)
        assert "s_setprio(3)" in s and "s_setprio(0)" in s
    assert "bool skewed" in s and "mid_barrier && g + 4" in s and "grp == 0 && (l == 4" in s
    return s


def build():
    from plenoctree_amd import build as b
    os.makedirs(OUT, exist_ok=True)
    src = open(os.path.join(CSRC, "mlp_kernels.hip")).read()
    patched = patch(src)
    if "--skew" in sys.argv:
        patched = patch_skew(patched)
    assert patched.count("STAMP(") >= 9, "anchors moved: update scripts/trace_mlp.py"
    for f in os.listdir(CSRC):
        if f.endswith(".h"):
            open(os.path.join(OUT, f), "w").write(open(os.path.join(CSRC, f)).read().replace('"../../include/', '"../../include/'))
    open(os.path.join(OUT, "mlp_kernels.hip"), "w").write(patched)
    objs = []
    for srcf in b.SOURCES:
        path = os.path.join(OUT if srcf == "mlp_kernels.hip" else CSRC, srcf)
        obj = os.path.join(OUT, srcf.replace(".hip", ".o"))
        cmd = ["/opt/rocm/bin/hipcc", *b.FLAGS, *b.SOURCE_FLAGS.get(srcf, []), "-I", CSRC, "-c", path, "-o", obj]
        subprocess.check_call(cmd)
        objs.append(obj)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs])
    print("built", LIB)


def run():
    import ctypes
    import torch
    from plenoctree_amd import _lib, ops
    from plenoctree_amd.nerf_sh.nerf import models
    lib = _lib.load()
    dev = torch.device("cuda:0")
    cfg = ops.make_cfg()
    flat = models.init_params(cfg)
    n = flat.numel() // 2
    pf, _ = ops.pack_weights(cfg, flat[n:].contiguous().to(dev), need_bwd=False)
    M = 4096 * 192 + 10000
    pts = torch.rand(M, 3, device=dev) * 4 - 2
    lib.pxo_debug_trace.restype = ctypes.c_int
    lib.pxo_debug_trace.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int, ctypes.c_int]
    for _ in range(3):
        ops.mlp_fwd(cfg, pf, pts, save=True)
    torch.cuda.synchronize()
    names = {0: "acc init / bias", 1: "GEMM", 2: "post-GEMM barrier wait", 3: "epilogue (LDS writes, mask)", 4: "tile-column stores issued",
             5: "post-epilogue barrier wait"}
    for which in (0, 1):
        buf = (ctypes.c_ulonglong * 4096)()
        nrec = lib.pxo_debug_trace(buf, which, 4096)
        recs = [((buf[i] >> 52) & 0xF, (buf[i] >> 48) & 0xF, buf[i] & 0xFFFFFFFFFFFF) for i in range(nrec)]
        # split into tiles at stamp 14
        tiles, cur = [], []
        for r in recs:
            if r[1] == 14 and cur:
                tiles.append(cur); cur = []
            cur.append(r)
        tiles = [t for t in tiles if len(t) == len(tiles[len(tiles) // 2])][1:]        # complete, steady-state tiles
        print(f"wave {4 * which}: {nrec} stamps, {len(tiles)} whole tiles")
        if not tiles:
            continue
        import collections
        acc = collections.defaultdict(list)
        for t in tiles:
            for (l0, k0, t0), (l1, k1, t1) in zip(t[:-1], t[1:]):
                acc[(l1, k1)].append(t1 - t0)
            acc[("tile", 0)].append(t[-1][2] - t[0][2])
        tot = sum(acc[("tile", 0)]) / len(tiles)
        print(f"  cycles per tile (first to last stamp): {tot:.0f}")
        per_kind = collections.defaultdict(float)
        for (l, k), v in sorted(acc.items(), key=lambda kv: (str(kv[0][0]), kv[0][1])):
            if l == "tile":
                continue
            m = sum(v) / len(v)
            per_kind[k] += m
            print(f"    layer {l:2} -> stamp {k:2}: {m:9.0f} cycles ({100 * m / tot:5.2f} %)")
        print("  by phase over the tile:")
        for k, m in sorted(per_kind.items()):
            nm = names.get(k, {14: "head GEMM + outputs (previous tile) ", 15: "encoding + barriers"}.get(k, str(k)))
            print(f"    {nm:40s} {m:9.0f} cycles ({100 * m / tot:5.2f} %)")


if __name__ == "__main__":
    {"build": build, "run": run}[sys.argv[1]]()
