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
    """Bucketed gradient all-reduce for data-parallel training of KVNET, overlapped with backward.

    Parameters that share storage (the feature CNN is registered twice) are reduced once.  Every bucket owns ONE
    persistent flat fp32 buffer and the parameters' `.grad` are views into it, so autograd accumulates straight into the
    message (no per-step torch.cat / scatter-back copies).  Buckets are filled in reverse registration order — the order
    backward produces gradients — and a bucket's all-reduce (sum, asynchronous) is launched from a post-accumulate hook
    once its last gradient has landed AND every earlier bucket has been launched, i.e. while backward is still running on
    the earlier layers; `__call__()` (after backward) launches whatever did not fire (parameters unused in the step
    contribute zeros), waits, and divides by the world size.
    Collective ORDER is the same on every rank by construction: buckets are only ever launched in index order 0, 1, 2, ...
    — a rank whose step leaves a sub-network unused (train() decides `valid_dpv(BVs_predict)` per rank: one rank may skip
    the K-Net while another runs it) simply stops launching from hooks at the first incomplete bucket and issues the rest
    from `__call__()`, in the same order as its peers; sizes are fixed at construction.  Few large buckets: xGMI is
    point-to-point and a ring all-reduce is per-link bound — 21.15 MB of fp32 gradient -> four messages at the default
    6 MB, so that the first three travel under the rest of backward.

    Use:  reducer.prepare(); loss.backward(); reducer(); optimizer.step()
    (`prepare` re-attaches the views — optimizers' zero_grad(set_to_none=True) drops them — and zeroes the buffers.)

    Gradient accumulation (BASELINE config 4: global batch 32 = 8 GPUs x 4 sequential N = 1 windows):
        reducer.prepare(accum_steps=4); four times { loss.backward() }; reducer(); optimizer.step()
    the four backward passes add into the same buckets, a bucket's collective starts (from the LAST pass's hooks) once all
    4 x len(bucket) gradients have landed, and `__call__` divides by 4 * world: the mean over all 32 windows.  With one rank
    the division by accum_steps still happens (no collective).

    `hold = True` keeps the hooks from launching anything (train_step.TrainGraph: forward + backward are captured into a
    hipGraph, whose replays run no hooks; every bucket is then launched from `__call__`, in index order as always).
    """

    def __init__(self, module, bucket_mb=6.0, overlap=True, always_collective=False):
        seen, self.params = set(), []
        for p in module.parameters():
            if p.requires_grad and p.data_ptr() not in seen:
                seen.add(p.data_ptr())
                self.params.append(p)
        limit = int(bucket_mb * 1024 * 1024 / 4)
        self.buckets, cur, n = [], [], 0
        for p in reversed(self.params):            # backward order: last layers first
            if cur and n + p.numel() > limit:
                self.buckets.append(cur)
                cur, n = [], 0
            cur.append(p)
            n += p.numel()
        if cur:
            self.buckets.append(cur)
        self.flat = [torch.zeros(sum(p.numel() for p in b), dtype=torch.float32, device=b[0].device) for b in self.buckets]
        self._views, self._bucket_of = {}, {}
        for bi, (b, flat) in enumerate(zip(self.buckets, self.flat)):
            off = 0
            for p in b:
                self._views[p] = flat[off:off + p.numel()].view_as(p)
                self._bucket_of[p] = bi
                off += p.numel()
        self._pending = [0] * len(self.buckets)
        self._work = [None] * len(self.buckets)
        self.overlap = overlap
        # issue the collectives on a ONE-rank process group as well (they are the identity there): lets a single-GPU box drive
        # the very code path of N > 1 — hook-launched asynchronous RCCL all-reduces during backward, the waits, the division —
        # through the real backend (tests/test_gpu_dist.py); off by default, one rank needs no collective
        self.always_collective = bool(always_collective)
        self.hold = False
        self._accum = 1
        self.launched_in_backward = 0              # buckets whose collective started from a hook in the last step
        if overlap and hasattr(torch.Tensor, "register_post_accumulate_grad_hook"):
            for p in self.params:
                p.register_post_accumulate_grad_hook(self._on_grad)
        self.prepare()

    @property
    def numel(self):
        return sum(p.numel() for p in self.params)

    def _active(self):
        return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or self.always_collective)

    def prepare(self, accum_steps=1):
        """Before the (first) backward: zero the messages and make every .grad a view into its bucket."""
        if accum_steps < 1:
            raise ValueError("accum_steps must be >= 1")
        self._accum = int(accum_steps)
        for flat in self.flat:
            flat.zero_()
        for p in self.params:
            p.grad = self._views[p]
        self._pending = [len(b) * self._accum for b in self.buckets]
        self._work = [None] * len(self.buckets)
        self._next = 0                             # first bucket not launched yet: launches happen in index order only
        self.launched_in_backward = 0
        self.launch_order = []                     # (bucket, "hook" | "call") of the last step, for tests / debugging

    def _launch(self, bi):
        if self._work[bi] is None and self._active():
            self._work[bi] = dist.all_reduce(self.flat[bi], op=dist.ReduceOp.SUM, async_op=True)

    def _on_grad(self, p):
        bi = self._bucket_of.get(p)
        if bi is None:
            return
        if p.grad is not self._views[p]:           # autograd replaced the tensor (first accumulation into a None grad)
            self._views[p].copy_(p.grad)
            p.grad = self._views[p]
        self._pending[bi] -= 1
        if self.overlap and not self.hold and self._active():
            # in index order only: a complete bucket behind an incomplete one waits (for __call__ at the latest), so that
            # every rank issues the same sequence of collectives whatever sub-networks its step used
            while self._next < len(self.buckets) and self._pending[self._next] == 0:
                self._launch(self._next)
                self.launch_order.append((self._next, "hook"))
                self.launched_in_backward += 1
                self._next += 1

    def __call__(self):
        if not self._active():
            if self._accum > 1:                    # one rank: the mean over the accumulated windows, no collective
                for p in self.params:
                    if p.grad is not None and p.grad is not self._views[p]:
                        self._views[p].copy_(p.grad)
                        p.grad = self._views[p]
                for flat in self.flat:
                    flat.div_(self._accum)
            return
        world = dist.get_world_size() * self._accum
        for p in self.params:                      # no hook support / a replaced grad tensor: fold it into the message
            if p.grad is not None and p.grad is not self._views[p] and self._work[self._bucket_of[p]] is None:
                self._views[p].copy_(p.grad)
                p.grad = self._views[p]
        for bi in range(self._next, len(self.buckets)):
            self._launch(bi)
            self.launch_order.append((bi, "call"))
        self._next = len(self.buckets)
        for bi, w in enumerate(self._work):
            w.wait()
            self.flat[bi].div_(world)


def graph_capture_mode():
    """capture_error_mode for torch.cuda.graph: with a process group alive, RCCL's watchdog / heartbeat threads may query events
    while THIS thread captures a hipGraph; under the default "global" mode any such call from another thread aborts the capture.
    "thread_local" confines the check to the capturing thread (what data-parallel training with captured steps needs)."""
    return "thread_local" if dist.is_available() and dist.is_initialized() else "global"
