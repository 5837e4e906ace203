"""Multi-GPU layer of the depth path: one process per GPU, torch.distributed (RCCL over xGMI on ROCm,
gloo on CPU for tests).

The reference uses single-process `torch.nn.DataParallel` (test_KVNet.py:163-164, train_KVNet.py:261-262):
per iteration it broadcasts 21 MB of weights, scatters the inputs, gathers 2 x 25 MB of outputs and
reduce-adds the gradients to GPU 0.  Here

  * inference shards by independent video stream — every stream carries its own DPV state, BatchNorm
    statistics are per sample, so replicas never talk: NO data-path collective (`shard_streams`);
  * training is data-parallel over windows with ONE collective per step, an all-reduce (sum, / world) of the
    flattened fp32 gradient (21.15 MB for the 5.29 M parameters) in a few large buckets — xGMI is
    point-to-point and ring all-reduce is per-link bound, so few big messages beat many small ones
    (`GradAllReduce`).  BatchNorm stays per replica, like DataParallel without SyncBN.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise the default process group from RANK / WORLD_SIZE / MASTER_* (torchrun).  Returns (rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        kw = {}
        if backend == "nccl":
            local = int(os.environ.get("LOCAL_RANK", "0"))
            torch.cuda.set_device(local)
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world


def shard_streams(n_streams, world, rank):
    """Round-robin assignment of independent video streams to ranks: stream s -> rank s % world."""
    if not 0 <= rank < world:
        raise ValueError("rank %d outside world %d" % (rank, world))
    return list(range(rank, n_streams, world))


def max_over_ranks(seconds, device=None):
    """Wall time of the slowest rank (the bench contract: barrier, time, MAX over ranks)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


class GradAllReduce:
    """Bucketed gradient all-reduce for data-parallel training of KVNET.

    Parameters that share storage (the feature CNN is registered twice) are reduced once.  Gradients are
    packed into contiguous fp32 buckets of `bucket_mb`, all-reduced (sum) asynchronously, divided by the world
    size and scattered back.  Parameters without a gradient (unused in the step) contribute zeros, so every
    rank issues the same collectives.
    """

    def __init__(self, module, bucket_mb=32.0):
        seen, self.params = set(), []
        for p in module.parameters():
            if p.requires_grad and p.data_ptr() not in seen:
                seen.add(p.data_ptr())
                self.params.append(p)
        limit = int(bucket_mb * 1024 * 1024 / 4)
        self.buckets, cur, n = [], [], 0
        for p in self.params:
            if cur and n + p.numel() > limit:
                self.buckets.append(cur)
                cur, n = [], 0
            cur.append(p)
            n += p.numel()
        if cur:
            self.buckets.append(cur)

    @property
    def numel(self):
        return sum(p.numel() for p in self.params)

    def __call__(self):
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return
        world = dist.get_world_size()
        work = []
        for bucket in self.buckets:
            flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in bucket])
            work.append((dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True), flat, bucket))
        for handle, flat, bucket in work:
            handle.wait()
            flat.div_(world)
            off = 0
            for p in bucket:
                g = flat[off:off + p.numel()].view_as(p)
                if p.grad is None:
                    p.grad = g.clone()
                else:
                    p.grad.copy_(g)
                off += p.numel()
