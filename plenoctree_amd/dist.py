"""One process per GPU over torch.distributed (backend "nccl" = RCCL over xGMI on ROCm; "gloo" for
the CPU tests).  Replaces the collectives XLA issues under jax.pmap in the reference:
lax.pmean(grad/stats) (nerf_sh/train.py:117-118) and lax.all_gather (nerf_sh/nerf/utils.py:703-706,
719-724)."""
import os

import torch


class Comm:
    def __init__(self, world=1, rank=0, local_rank=0, backend=None):
        self.world, self.rank, self.local_rank, self.backend = world, rank, local_rank, backend

    @property
    def is_dist(self):
        return self.world > 1

    def all_reduce_sum(self, t):
        """In-place sum over ranks (one call on the whole flat gradient arena)."""
        if self.is_dist:
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.SUM)
        return t

    def all_reduce_max(self, t):
        """In-place maximum over ranks (per-voxel weight mask accumulated over camera shards)."""
        if self.is_dist:
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        return t

    def all_gather_cat(self, t):
        """Concatenate equally-shaped per-rank tensors along dim 0, in rank order."""
        if not self.is_dist:
            return t
        out = torch.empty((self.world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        torch.distributed.all_gather_into_tensor(out, t.contiguous())
        return out

    def all_gather_into(self, out, t):
        """all_gather of equal-size contiguous `t` straight into the flat tensor `out` (world * t.numel())."""
        if not self.is_dist:
            out.copy_(t.reshape(-1))
            return out
        src = t.reshape(-1)
        if self.backend != "nccl":        # RCCL gathers in place when `t` is this rank's slice of `out`; gloo gets a copy
            src = src.clone()
        torch.distributed.all_gather_into_tensor(out, src)
        return out

    def barrier(self):
        if self.is_dist:
            torch.distributed.barrier()

    def shutdown(self):
        if self.is_dist and torch.distributed.is_initialized():
            torch.distributed.destroy_process_group()


def init_from_env(backend=None, device=None):
    """RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the launcher (torch.distributed.run)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not torch.distributed.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        torch.distributed.init_process_group(backend, rank=rank, world_size=world, **kw)
    return Comm(world, rank, local_rank, backend)


def slab_range(reso, world, rank):
    """x-slab [x0,x1) of a reso^3 grid owned by `rank` (contiguous, sizes differ by at most 1)."""
    base, rem = divmod(reso, world)
    x0 = rank * base + min(rank, rem)
    return x0, x0 + base + (1 if rank < rem else 0)
