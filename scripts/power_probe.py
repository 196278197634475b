"""Board power and shader clock while one hot kernel runs back to back (rocm-smi sampled from a side thread).
    python scripts/power_probe.py fwd|bwd|wgrad|x3 [seconds]"""
import os, subprocess, sys, threading, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from plenoctree_amd import ops
from plenoctree_amd.nerf_sh.nerf import models
which = sys.argv[1] if len(sys.argv) > 1 else "fwd"
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
dev = torch.device("cuda:0")
cfg = ops.make_cfg(mlp_precision=1 if which == "x3" else 0)
flat = models.init_params(cfg)
n = flat.numel() // 2
pf, pb = ops.pack_weights(cfg, flat[n:].contiguous().to(dev), need_bwd=which != "x3")
M = 4096 * 192
pts = torch.rand(M, 3, device=dev) * 4 - 2
samples, stop = [], False
def sampler():
    while not stop:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True).stdout
        pw = [l for l in out.splitlines() if "Power" in l and "W" in l]
        ck = [l for l in out.splitlines() if "sclk" in l]
        samples.append((time.time(), pw[:1], ck[:1]))
        time.sleep(0.3)
if which in ("bwd", "wgrad"):
    _, _, (acts, enc, mask) = ops.mlp_fwd(cfg, pf, pts, save=True)
    d_rgb = torch.randn(M, 48, device=dev) * 0.1; d_sig = torch.randn(M, device=dev) * 0.1
    dz, dbias = ops.mlp_bwd_data(cfg, pb, d_rgb, d_sig, mask)
def once():
    if which == "fwd":
        ops.mlp_fwd(cfg, pf, pts, save=True)
    elif which == "x3":
        ops.mlp_fwd(cfg, pf, pts, save=False)
    elif which == "bwd":
        ops.mlp_bwd_data(cfg, pb, d_rgb, d_sig, mask)
    else:
        ops.mlp_bwd_weights(cfg, acts, enc, dz, d_rgb, d_sig, dbias)
once(); torch.cuda.synchronize()
th = threading.Thread(target=sampler); th.start()
t0 = time.time(); it = 0
while time.time() - t0 < secs:
    for _ in range(10):
        once()
    torch.cuda.synchronize(); it += 10
dt = time.time() - t0
stop = True; th.join()
print(which, "iterations", it, "ms/iter", 1e3 * dt / it)
for t, pw, ck in samples:
    print(f"{t - t0:6.2f}s", (pw[0].split(":")[-1].strip() if pw else "?"), "|", (ck[0].split(":", 2)[-1].strip() if ck else "?"))
