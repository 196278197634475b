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

    def _gather_flat(self, out, src):
        """all_gather of equal-size contiguous `src` into flat `out`.  gloo has no all_gather_into_tensor for device
        tensors (CPU tests and the shared-GPU functional run use it): list form there."""
        if self.backend == "gloo" and src.is_cuda:
            parts = [torch.empty_like(src) for _ in range(self.world)]
            torch.distributed.all_gather(parts, src)
            torch.cat([p.reshape(-1) for p in parts], out=out.reshape(-1))
        else:
            torch.distributed.all_gather_into_tensor(out, src)

    def all_gather_cat(self, t):
        """Concatenate equally-shaped per-rank tensors along dim 0, in rank order."""
        if not self.is_dist:
            return t
        out = torch.empty((self.world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        self._gather_flat(out, t.contiguous())
        return out

    def all_gather_into(self, out, t):
        """all_gather of equal-size contiguous `t` straight into the flat tensor `out` (world * t.numel())."""
        if not self.is_dist:
            out.copy_(t.reshape(-1))
            return out
        src = t.reshape(-1)
        if self.backend != "nccl":        # RCCL gathers in place when `t` is this rank's slice of `out`; gloo gets a copy
            src = src.clone()
        self._gather_flat(out, src)
        return out

    def barrier(self):
        if self.is_dist:
            torch.distributed.barrier()

    def shutdown(self):
        if self.is_dist and torch.distributed.is_initialized():
            torch.distributed.destroy_process_group()


_SIDE_STREAMS = {}


def exchange_stream(device, priority=-1):
    """The high-priority stream the gradient exchange hangs its first bucket on; one per device and priority, created once
    (bench.py creates it together with the process group, before any kernel runs: created later, after seconds of compute,
    the same stream cost the 512-ray step 2 %, profiles/r04h_late_group.txt)."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), priority)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=device, priority=priority)
    return _SIDE_STREAMS[key]


class GradReducer:
    """lax.pmean(grad) + lax.pmean(stats) (nerf_sh/train.py:117-118) as TWO sum-all-reduces per step instead of one
    after the backward pass: MLP_0's half of the gradient arena is final once the coarse level has been reversed -- a
    quarter into the step (pxo_train_fwd_bwd_bucketed) -- and is reduced on a side stream under the fine level; MLP_1's
    half with the 6 stats in its tail follows when the step's kernels are through.  The caller's stream waits for both
    before Adam, so the reduced values -- and the bit-identical replicas -- are those of the single call.

    On RCCL both collectives run on the process group's own stream in issue order (bucket 0, bucket 1); what the side
    stream adds is the *dependency*: bucket 0 waits for the library's `grads0_ready` event only, not for the kernels
    queued behind it.  With gloo (CPU tests) the two calls are simply made in that order.  `all_reduce` replaces the
    collective (tests count the calls through it)."""

    def __init__(self, comm, device=None, all_reduce=None, force=False, side_priority=-1):
        self.comm = comm
        self.active = comm.is_dist or force or all_reduce is not None
        self.all_reduce = all_reduce
        self.on_gpu = device is not None and device.type == "cuda"
        self.event = self.side = self.bucket0_done = None
        self.record_timing = False            # tests: make `bucket0_done` a timing event
        if self.active and self.on_gpu:
            from . import ops
            self.event = ops.Event()
            # HIGH priority, and the process group's stream too (nccl_options below): HIP multiplexes streams onto a few
            # hardware queues, and a collective whose queue also carries the step's kernels is dispatched behind all of
            # them -- measured (scripts/overlap_probe.py, profiles/r04h_overlap_probe.txt): with normal-priority streams
            # bucket 0 is done 0.03 ms AFTER the step's last kernel, with high-priority ones 9.1 ms BEFORE it (2048 rays)
            self.side = exchange_stream(device, side_priority)

    def ready_event(self):
        """What pxo_train_fwd_bwd_bucketed records when bucket 0 is final (None: no overlap, e.g. on the CPU)."""
        return self.event

    def _reduce(self, t, async_op):
        if self.all_reduce is not None:
            self.all_reduce(t)
            return None
        return torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.SUM, async_op=async_op)

    def reduce(self, bucket0, bucket1):
        """Sum both buckets over the ranks; on return the current stream is ordered after both collectives."""
        if not self.active:
            return
        if self.side is None:
            self._reduce(bucket0, False)
            self._reduce(bucket1, False)
            return
        with torch.cuda.stream(self.side):
            self.event.wait(self.side)              # ... and nothing else of this step
            w0 = self._reduce(bucket0, True)
            if self.record_timing:                  # tests: when was bucket 0 done, on the device's clock
                if w0 is not None:
                    w0.wait()
                self.bucket0_done = torch.cuda.Event(enable_timing=True)
                self.bucket0_done.record(self.side)
        if w0 is None:
            # an injected `all_reduce` (tests, scripts/contention_probe.py) returns no Work handle: whatever it enqueued on
            # the side stream is joined explicitly, or Adam could read bucket 0 before its reduction has finished
            torch.cuda.current_stream().wait_stream(self.side)
        w1 = self._reduce(bucket1, True)            # ordered after everything queued on the current stream
        # the caller's stream waits for both collectives directly (Work.wait orders the CURRENT stream behind the work):
        # no event hop through the side stream on the way back
        if w0 is not None:
            w0.wait()
        if w1 is not None:
            w1.wait()


def nccl_options():
    """ProcessGroupNCCL options for every RCCL group of this package: the collective stream is a HIGH-priority stream, so that
    it gets a hardware queue of its own class and bucket 0's all-reduce can run under the fine level (GradReducer)."""
    opts = torch.distributed.ProcessGroupNCCL.Options()
    opts.is_high_priority_stream = True
    return opts


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
        if backend == "nccl":
            kw["pg_options"] = nccl_options()
        torch.distributed.init_process_group(backend, rank=rank, world_size=world, **kw)
        if backend == "nccl" and torch.cuda.is_available():
            exchange_stream(torch.device("cuda", local_rank))     # early, see exchange_stream
    return Comm(world, rank, local_rank, backend)


def single_host(world=None):
    """Do all ranks of this job sit on ONE host?  torch.distributed.run exports LOCAL_WORLD_SIZE (ranks on this node): equal to
    WORLD_SIZE on one node.  Without a launcher (mp.spawn in tests) a loopback MASTER_ADDR says the same."""
    world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else world
    if world <= 1:
        return True
    lws = os.environ.get("LOCAL_WORLD_SIZE")
    if lws is not None:
        return int(lws) == world
    return os.environ.get("MASTER_ADDR", "127.0.0.1") in ("127.0.0.1", "localhost", "::1")


def per_host_image(flag, world=None):
    """Resolve --per_host_image (true / false / auto).  auto = what the reference does on the machine at hand: ranks that share
    a host are its `jax.local_devices()` -- ONE image per step, its batch_size pixels drawn once and sharded over them
    (nerf_sh/nerf/datasets.py:159-166 + nerf_sh/nerf/utils.py:518-522); ranks on different hosts each draw their own image
    (np.random.seed(20201473 + jax.host_id()), nerf_sh/train.py:128).  Hybrid layouts (several hosts x several ranks) keep the
    multi-host sampler unless the flag says otherwise."""
    world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else world
    if isinstance(flag, str):
        f = flag.lower()
        if f in ("auto", ""):
            return world > 1 and single_host(world)
        return f in ("1", "true", "yes", "y")
    if flag is None:
        return world > 1 and single_host(world)
    return bool(flag)


def slab_range(reso, world, rank):
    """x-slab [x0,x1) of a reso^3 grid owned by `rank` (contiguous, sizes differ by at most 1)."""
    base, rem = divmod(reso, world)
    x0 = rank * base + min(rank, rem)
    return x0, x0 + base + (1 if rank < rem else 0)
