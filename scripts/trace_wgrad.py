"""Run the cycle-stamped 256x256 wgrad kernel (plenoctree_amd/libplenoctree_hip_wtrace.so, `build.py --trace`) and print
the per-phase cycles of wave 0 of workgroup 0 over a few steady-state chunks."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PXO_LIB"] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "plenoctree_amd", "libplenoctree_hip_wtrace.so")
from plenoctree_amd import _lib, ops
lib = _lib.load()
dev = torch.device("cuda:0")
cfg = ops.make_cfg()
M = 4096 * 192 + 10000
acts = torch.randn(8, M, 256, device=dev); enc = torch.randn(M, 64, device=dev); dz = torch.randn(8, M, 256, device=dev)
d_rgb = torch.randn(M, 48, device=dev); d_sig = torch.randn(M, device=dev)
dbias = torch.zeros(ops.dbias_partial_bytes(M) // 4, device=dev)
lib.pxo_debug_wtrace.restype = ctypes.c_int
lib.pxo_debug_wtrace.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int, ctypes.c_int]
buf = (ctypes.c_ulonglong * 2048)()
ops.mlp_bwd_weights(cfg, acts, enc, dz, d_rgb, d_sig, dbias)
lib.pxo_debug_wtrace(buf, 2048, 1)
ops.mlp_bwd_weights(cfg, acts, enc, dz, d_rgb, d_sig, dbias)
n = lib.pxo_debug_wtrace(buf, 2048, 1)
recs = [(buf[i] >> 48, buf[i] & 0xFFFFFFFFFFFF) for i in range(n)]
print("records", n)
prev = None
for tag, t in recs[: 8 * 13]:          # first launch (layer 1): 8 chunks x 13 stamps
    print(f"{tag:3d} +{(t - prev) if prev is not None else 0:8d}")
    prev = t
