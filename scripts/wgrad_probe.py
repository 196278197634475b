"""Is the main wgrad GEMM limited by HBM streaming?  Time pxo_mlp_bwd_weights' 256x256 launches for
several M (operands 2 x M x 1 KiB): small M stays in the 256 MB Infinity Cache across repetitions."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from plenoctree_amd import ops
dev = torch.device("cuda:0")
cfg = ops.make_cfg()
for M in (32768, 65536, 262144, 796544):
    acts = torch.randn(8, M, 256, device=dev); dz = torch.randn(8, M, 256, device=dev)
    enc = torch.randn(M, 64, device=dev); drgb = torch.randn(M, 48, device=dev); dsig = torch.randn(M, device=dev)
    dbias = torch.zeros(ops.dbias_partial_bytes(M) // 4, device=dev)
    for rep in range(3):
        ops.mlp_bwd_weights(cfg, acts, enc, dz, drgb, dsig, dbias)
    torch.cuda.synchronize()
    ops.profile_enable(True)
    for rep in range(5):
        ops.mlp_bwd_weights(cfg, acts, enc, dz, drgb, dsig, dbias)
    torch.cuda.synchronize()
    n, ms, rows = ops.profile_read(2)
    ops.profile_enable(False)
    print(f"M={M:7d}: {n} launches, {ms / n * 1e3:8.1f} us avg, {rows * 131072 / (ms * 1e-3) / 1e12:6.1f} TFLOP/s "
          f"(operands per launch {2 * M * 1024 / 2**20:.0f} MiB, all 8 layers {16 * M * 1024 / 2**20:.0f} MiB)")
    del acts, dz
