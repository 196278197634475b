"""Run the cycle-stamped forward kernel (PXO_LIB=..._trace_*.so) and print per-phase cycles of wave 0."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
buf = (ctypes.c_ulonglong * 512)()
lib.pxo_debug_trace.restype = ctypes.c_int
lib.pxo_debug_trace.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int, ctypes.c_int]
ops.mlp_fwd(cfg, pf, pts, save=True)            # warm
lib.pxo_debug_trace(buf, 512, 1)
ops.mlp_fwd(cfg, pf, pts, save=True)
nrec = lib.pxo_debug_trace(buf, 512, 1)
recs = [(buf[i] >> 48, buf[i] & 0xFFFFFFFFFFFF) for i in range(nrec)]
print("tile rows", lib.pxo_tile_rows(), "records", nrec)
prev = None
for tag, t in recs:
    print(f"{tag:3d} +{(t - prev) if prev is not None else 0:8d}")
    prev = t
