#!/usr/bin/env python
"""Cycle stamps of the bf16x6 forward kernel (csrc/mlp_x6_kernels.hip), from an instrumented COPY of the source (the shipped
source carries no trace hooks; same method as scripts/trace_mlp.py): `python scripts/trace_x6.py build` patches the file into
scripts/trace_build/ and links plenoctree_amd/libplenoctree_hip_trace.so; `python scripts/trace_x6.py run` (GPU box, with
PXO_ALLOW_VARIANT=1 PXO_LIB=plenoctree_amd/libplenoctree_hip_trace.so) runs the training forward on 4096 x 192 + 10000 rows
and prints the average cycles per phase of waves 0 and 4 (the two priorities of SIMD 0) of workgroup 0.

Stamps (s_memtime): 14 sub-tile start, 15 after the encoding barrier, 13 after the heads; per layer
  0 before the GEMM   1 after the GEMM(s)   2 epilogue registers / global stores done   3 after the planes-consumed barrier
  4 plane writes issued   5 after the planes-written barrier
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
// ---- injected by scripts/trace_x6.py ---------------------------------------------------------
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


def rep(s, old, new):
    assert s.count(old) == 1, f"anchor moved ({s.count(old)} matches): {old[:80]!r}"
    return s.replace(old, new, 1)


def patch(s):
    s = s.replace("namespace pxo {\n", "namespace pxo {\n" + PREAMBLE, 1)
    s = rep(s, "float* __restrict__ enc_out, uint32_t* __restrict__ mask_sub, int tid, int wave) {",
            "float* __restrict__ enc_out, uint32_t* __restrict__ mask_sub, int tid, int wave,\n"
            "                                               unsigned long long* s_trace, int& s_tn, bool trace_on) {")
    s = rep(s, "  lds_barrier6();   // the previous sub-tile's head GEMM has consumed the planes\n  uint4 enc_keep[3];",
            "  { const int l = 15; STAMP(14); }\n  lds_barrier6();   // the previous sub-tile's head GEMM has consumed the planes\n  uint4 enc_keep[3];")
    s = rep(s, "  posenc_tile_x6<SAVE, GRID>(planes, pts, grid, row0, M, tid, enc_out, enc_keep);\n  lds_barrier6();\n",
            "  posenc_tile_x6<SAVE, GRID>(planes, pts, grid, row0, M, tid, enc_out, enc_keep);\n  lds_barrier6();\n  { const int l = 15; STAMP(15); }\n")
    s = rep(s, "    const int nkg = l == 0 ? 4 : 16;\n    gemm_x6<kYRB, true>(", "    const int nkg = l == 0 ? 4 : 16;\n    STAMP(0);\n    gemm_x6<kYRB, true>(")
    s = rep(s, "    // Epilogue, first half -- registers and global memory only, so it needs no barrier:",
            "    STAMP(1);\n    // Epilogue, first half -- registers and global memory only, so it needs no barrier:")
    s = rep(s, "    lds_barrier6();  // every wave has consumed the input planes\n",
            "    STAMP(2);\n    lds_barrier6();  // every wave has consumed the input planes\n    STAMP(3);\n")
    s = rep(s, "      for (int p2 = 0; p2 < 2; ++p2) store_chunk(planes, pw, pw2, r * 32 * kLDB + 16 * p2, pc[r][p2]);\n    lds_barrier6();\n  }\n\n  // heads",
            "      for (int p2 = 0; p2 < 2; ++p2) store_chunk(planes, pw, pw2, r * 32 * kLDB + 16 * p2, pc[r][p2]);\n    STAMP(4);\n    lds_barrier6();\n    STAMP(5);\n  }\n\n  // heads")
    # heads: after the sliced GEMM (6), after the partial sums are staged (7), after the outputs are written (13)
    s = rep(s, "    lds_barrier6();                                      // every wave is through with the planes\n",
            "    { const int l = 15; STAMP(6); }\n    lds_barrier6();                                      // every wave is through with the planes\n")
    s = rep(s, "    const float* hb = bias + 8 * kW;\n    const int col0 =", "    { const int l = 15; STAMP(7); }\n    const float* hb = bias + 8 * kW;\n    const int col0 =")
    s = rep(s, "      raw_sigma[row0 + tid] = v + hb[C];\n    }\n  }\n}\n\ntemplate <int NHB, bool SAVE, bool RGB, bool GRID, bool DYN>",
            "      raw_sigma[row0 + tid] = v + hb[C];\n    }\n  }\n  { const int l = 15; STAMP(13); }\n}\n\ntemplate <int NHB, bool SAVE, bool RGB, bool GRID, bool DYN>")
    # kernel: trace buffers, pass-through, flush
    s = rep(s, "  __shared__ int s_next[4];\n  const int tid = threadIdx.x;\n  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);\n  const float* __restrict__ bias = pk + x6_fwd_bias_off(deg);",
            "  __shared__ int s_next[4];\n  __shared__ unsigned long long s_trace_all[2][1536];\n  const int tid = threadIdx.x;\n  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);\n"
            "  const bool trace_on = blockIdx.x == 0 && (tid & 63) == 0 && (wave == 0 || wave == 4);\n  unsigned long long* s_trace = s_trace_all[wave >> 2];\n  int s_tn = 0;\n"
            "  const float* __restrict__ bias = pk + x6_fwd_bias_off(deg);")
    assert s.count("mslot + sub, tid, wave);") == 1 and s.count("raw_rgb, raw_sigma, acts, enc_out, mslot, tid, wave);") == 1
    s = s.replace("mslot + sub, tid, wave);", "mslot + sub, tid, wave, s_trace, s_tn, trace_on);")
    s = s.replace("raw_rgb, raw_sigma, acts, enc_out, mslot, tid, wave);", "raw_rgb, raw_sigma, acts, enc_out, mslot, tid, wave, s_trace, s_tn, false);")
    s = rep(s, "      run(slot);\n      slot = tk.take(tid, ticket);\n    }\n  }\n}\n\nstatic unsigned x6_grid(int64_t M) {",
            "      run(slot);\n      slot = tk.take(tid, ticket);\n    }\n  }\n  if (trace_on) {\n    const int w = wave >> 2;\n    for (int i = 0; i < s_tn; ++i) g_trace[w][i] = s_trace[i];\n    g_trace_n[w] = s_tn;\n  }\n}\n\nstatic unsigned x6_grid(int64_t M) {")
    s = s.replace("}  // namespace pxo", '''}  // namespace pxo
extern "C" int pxo_debug_trace(unsigned long long* out, int which, int cap) {
  int n = 0;
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(&n, HIP_SYMBOL(pxo::g_trace_n), sizeof(int), which * sizeof(int));
  if (n > cap) n = cap;
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(pxo::g_trace), sizeof(unsigned long long) * n, (size_t)which * 4096 * sizeof(unsigned long long));
  return n;
}''', 1)
    # DIAGNOSTICS (never shipped): where does the GEMM phase lose its cycles?
    if "--w-l1" in sys.argv:       # every weight fragment from one 3 KB region per wave: L1 hits, no L2 latency / bandwidth
        s = rep(s, "  const int idx = wu + kg * kg_stride;\n", "  const int idx = (wu & 0x7ff) + 0 * kg * kg_stride;\n")
    if "--x-once" in sys.argv:     # activation fragments read from ONE k-group: a quarter... the same LDS traffic, no new addresses
        s = rep(s, "    for (int i = 0; i < 3; ++i) x.p[r][i] = *reinterpret_cast<const bf16x8*>(xp + i * kPlane + r * 32 * kLDB + kg * 16);",
                "    for (int i = 0; i < 3; ++i) x.p[r][i] = *reinterpret_cast<const bf16x8*>(xp + i * kPlane + r * 32 * kLDB + 0 * kg * 16);")
    if "--no-store" in sys.argv:   # the float32 copy of the activations is not stored (nothing else changes)
        s = rep(s, "          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), out, st_voff, (r * 32 * kW + 8 * q) * 4, 0);\n        } else {",
                "          asm volatile(\"\" :: \"v\"(v));\n        } else {")
    if "--rr4" in sys.argv:        # the 12 MFMAs of a k-group round-robin over FOUR accumulators (distance 4 instead of 2; wrong numbers)
        s = rep(s, """  PXO_X6_MFMA(hi, 0, 0);
  PXO_X6_MFMA(lo, 2, 0);
  PXO_X6_MFMA(lo, 0, 2);
  PXO_X6_MFMA(lo, 1, 1);
  PXO_X6_MFMA(lo, 1, 0);
  PXO_X6_MFMA(lo, 0, 1);
}""", """  PXO_X6_MFMA(hi, 0, 0);
  PXO_X6_MFMA(lo, 2, 0);
  PXO_X6_MFMA(hi, 0, 2);
  PXO_X6_MFMA(lo, 1, 1);
  PXO_X6_MFMA(hi, 1, 0);
  PXO_X6_MFMA(lo, 0, 1);
}""")
    if "--no-x" in sys.argv:       # no LDS reads inside the loop at all (fragments of k-group 0 reused)
        s = s.replace("    load_x6<RBN>(xp, g + 1, x1);\n", "    if (g == 0) load_x6<RBN>(xp, 1, x1);\n")
        s = s.replace("    load_x6<RBN>(xp, g + 2, x0);\n", "")
        s = s.replace("    load_x6<RBN>(xp, g + 3, x1);\n", "")
        s = s.replace("    load_x6<RBN>(xp, (g + 4 < kgroups ? g + 4 : kgroups - 1), x0);\n", "")
    return s


def build():
    from plenoctree_amd import build as b
    os.makedirs(OUT, exist_ok=True)
    patched = patch(open(os.path.join(CSRC, "mlp_x6_kernels.hip")).read())
    assert patched.count("STAMP(") >= 10
    open(os.path.join(OUT, "mlp_x6_kernels.hip"), "w").write(patched)
    objs, procs = [], []
    for srcf in b.SOURCES:
        path = os.path.join(OUT if srcf == "mlp_x6_kernels.hip" else CSRC, srcf)
        obj = os.path.join(OUT, srcf.replace(".hip", ".o"))
        cmd = ["/opt/rocm/bin/hipcc", *b.FLAGS, *b.SOURCE_FLAGS.get(srcf, []), "-I", CSRC, "-c", path, "-o", obj]
        procs.append(subprocess.Popen(cmd))
        objs.append(obj)
    for p in procs:
        assert p.wait() == 0
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs])
    print("built", LIB)


def run():
    import collections
    import ctypes
    import torch
    from plenoctree_amd import _lib, ops
    from plenoctree_amd.nerf_sh.nerf import models
    lib = _lib.load()
    dev = torch.device("cuda:0")
    cfg = ops.make_cfg(mlp_precision=2)
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
    names = {0: "bias pattern / loop set-up", 1: "GEMM", 2: "epilogue: registers + global stores", 3: "planes-consumed barrier wait",
             4: "plane writes issued", 5: "planes-written barrier wait", 6: "head GEMM (K slice)", 7: "head partial sums staged (2 barriers)", 13: "head sums + outputs", 14: "(gap to the next sub-tile)",
             15: "encoding + barriers"}
    for which in (0, 1):
        buf = (ctypes.c_ulonglong * 4096)()
        nrec = lib.pxo_debug_trace(buf, which, 4096)
        recs = [((buf[i] >> 52) & 0xF, (buf[i] >> 48) & 0xF, buf[i] & 0xFFFFFFFFFFFF) for i in range(nrec)]
        tiles, cur = [], []
        for r in recs:
            if r[1] == 14 and cur:
                tiles.append(cur); cur = []
            cur.append(r)
        tiles = [t for t in tiles if len(t) == len(tiles[len(tiles) // 2])][1:]        # complete, steady-state sub-tiles
        print(f"wave {4 * which}: {nrec} stamps, {len(tiles)} whole sub-tiles")
        if not tiles:
            continue
        acc = collections.defaultdict(list)
        for t in tiles:
            for (l0, k0, t0), (l1, k1, t1) in zip(t[:-1], t[1:]):
                acc[(l1, k1)].append(t1 - t0)
            acc[("tile", 0)].append(t[-1][2] - t[0][2])
        tot = sum(acc[("tile", 0)]) / len(tiles)
        print(f"  cycles per sub-tile (first to last stamp): {tot:.0f}")
        per_kind = collections.defaultdict(float)
        for (l, k), v in sorted(acc.items(), key=lambda kv: (str(kv[0][0]), kv[0][1])):
            if l == "tile":
                continue
            m = sum(v) / len(v)
            per_kind[k] += m
            print(f"    layer {l:2} -> stamp {k:2}: {m:9.0f} cycles ({100 * m / tot:5.2f} %)")
        print("  by phase over the sub-tile:")
        for k, m in sorted(per_kind.items()):
            print(f"    {names.get(k, str(k)):40s} {m:9.0f} cycles ({100 * m / tot:5.2f} %)")


if __name__ == "__main__":
    {"build": build, "run": run}[sys.argv[1]]()
