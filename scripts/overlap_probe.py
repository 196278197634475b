#!/usr/bin/env python
"""Does bucket 0's all-reduce really run under the fine level?  Single-rank RCCL group; prints how long before the step's
last kernel the collective of MLP_0's half was complete (positive = overlapped).  HIP multiplexes streams onto a few hardware
queues; two streams that share one dispatch in enqueue order, so the answer depends on which queue the process group's
stream landed on -- run with PXO_PROBE_HIPRIO=0/1 (normal / high priority for the reducer's side stream and the process group's stream) and PXO_PROBE_WARM=n
(n extra streams created first, to shift the round-robin)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
hiprio = os.environ.get("PXO_PROBE_HIPRIO", "1") == "1"
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    from oracle import nerf_oracle as O
    from _helpers import make_params, make_rays, pxo_cfg
    from plenoctree_amd import dist as pdist, ops
    from plenoctree_amd.nerf_sh.nerf import models
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    keep = [torch.cuda.Stream(dev) for _ in range(int(os.environ.get("PXO_PROBE_WARM", "0")))]
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
    from plenoctree_amd import dist as pdist0
    kw = {"pg_options": pdist0.nccl_options()} if hiprio else {}
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev, **kw)
    cfg = O.Cfg(sparsity_npoints=1000)
    pcfg = pxo_cfg(ops, cfg)
    B = 2048
    rays = [r.to(dev) for r in make_rays(B)]
    px = torch.rand(B, 3, device=dev)
    state = models.TrainState(pcfg, make_params(cfg).to(dev))
    model = models.NerfModel(pcfg)
    red = pdist.GradReducer(pdist.Comm(1, 0, 0, "nccl"), dev, force=True, side_priority=-1 if hiprio else 0)
    red.record_timing = True
    batch = {"rays": type(make_rays(1))(*rays), "pixels": px}
    for step in range(3):
        models.train_step(model, state, batch, 5e-4, randomized=True, seed=step, world_size=1, reducer=red)
    leads = []
    for step in range(3, 8):
        ws = state.workspace(ops.train_workspace_bytes(pcfg, B))
        end = torch.cuda.Event(enable_timing=True)
        ops.train_fwd_bwd(pcfg, state.params, state.packed, *rays, px, state.grads, state.stats, ws, randomized=True, seed=step,
                          grads0_ready=red.ready_event())
        end.record()
        red.reduce(state.bucket0, state.bucket1)
        torch.cuda.synchronize()
        leads.append(red.bucket0_done.elapsed_time(end))
    print(f"hiprio={int(hiprio)} warm={len(keep)} side_priority={red.side.priority}: bucket 0 done "
          + ", ".join(f"{x:.3f}" for x in leads) + " ms before the end of the step's kernels")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
