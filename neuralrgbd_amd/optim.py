"""Adam for the training loop on ONE hand-written multi-tensor kernel (csrc/optim.hip).

`FusedAdam` is a drop-in for the reference's `optim.Adam(model_KVnet.parameters(), lr=..., betas=(.9, .999))`
(train_KVNet.py:228-232; stepped at train_utils/train_KVNet.py:153): same constructor arguments (no amsgrad), same per-parameter
state names (`step`, `exp_avg`, `exp_avg_sq`) so checkpoints interchange with torch.optim.Adam, same arithmetic
(`torch.optim.adam._single_tensor_adam`, fp32).  What differs is the execution: the update of all 459 tensors is ~10 launches of
one kernel whose tensor pointers travel in the kernel arguments — no `_foreach_` slabs, no host synchronisation, capturable into a
hipGraph as it is (the step counters live on the device).  CUDA-device fp32 parameters only; anything else raises."""
import ctypes

import torch

from . import _lib


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, maximize=False):
        if lr < 0.0 or eps < 0.0 or weight_decay < 0.0 or not (0.0 <= betas[0] < 1.0 and 0.0 <= betas[1] < 1.0):
            raise ValueError("FusedAdam: invalid hyper-parameters lr=%r betas=%r eps=%r weight_decay=%r" % (lr, betas, eps, weight_decay))
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, maximize=maximize))
        self._tables = {}

    def _state_of(self, p):
        st = self.state[p]
        if not st:
            st["step"] = torch.zeros((), dtype=torch.float32, device=p.device)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        elif not torch.is_tensor(st["step"]):       # a checkpoint of the reference's torch era (< 1.12): `step` is a Python int
            st["step"] = torch.tensor(float(st["step"]), dtype=torch.float32, device=p.device)
        elif not (st["step"].is_cuda and st["step"].dtype == torch.float32):     # a torch.optim.Adam checkpoint: host-side counter
            st["step"] = st["step"].detach().to(device=p.device, dtype=torch.float32).reshape(())
        return st

    def load_state_dict(self, state_dict):
        """torch.optim.Adam checkpoints load as they are (train_KVNet.py:347 saves `optimizer.state_dict()`), except AMSGrad
        ones: this kernel keeps no running maximum of the second moment, and dropping it silently would change the training."""
        for group in state_dict.get("param_groups", ()):       # refuse BEFORE any state is replaced
            if group.get("amsgrad"):
                raise _lib.NrgbdError("FusedAdam: the loaded param_group has amsgrad=True; this optimizer has no AMSGrad form")
        super().load_state_dict(state_dict)
        # a checkpoint of the reference's torch era (< 1.12) carries only lr / betas / eps / weight_decay / amsgrad per group:
        # super() REPLACES the groups with the loaded ones, so the keys step() reads (`maximize`) are filled from the defaults
        for group in self.param_groups:
            for k, v in self.defaults.items():
                group.setdefault(k, v)
        self._tables = {}                                       # the moments are new tensors: the pointer tables are stale

    def __setstate__(self, state):
        super().__setstate__(state)
        for group in self.param_groups:
            for k, v in self.defaults.items():
                group.setdefault(k, v)
        self._tables = {}

    def mark_updated(self):
        """Advance the version counter of every parameter this optimizer owns.  The kernel writes through raw pointers, which
        autograd's version counters do not see; the inference-side caches (nets.py: packed weight streams, the clamped-FMA unit)
        are keyed on them.  step() calls this itself; a hipGraph REPLAY of a captured step() runs no Python, so whoever replays
        calls it (train_step.TrainGraph does)."""
        for group in self.param_groups:
            for p in group["params"]:
                if p.requires_grad:
                    torch._C._increment_version(p)

    def init_state(self):
        """Create the state of every parameter now (before a hipGraph capture: a state created inside a capture would become
        graph nodes that reset the moments at every replay)."""
        for group in self.param_groups:
            for p in group["params"]:
                if p.requires_grad:
                    self._state_of(p)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        for gi, group in enumerate(self.param_groups):
            ptrs, key = [], []
            for p in group["params"]:
                g = p.grad
                if g is None:
                    continue
                if not (p.is_cuda and p.dtype == torch.float32 and g.dtype == torch.float32 and not g.is_sparse):
                    raise _lib.NrgbdError("FusedAdam: fp32 dense parameters on the GPU only (got %s %s)" % (p.device, p.dtype))
                if not p.is_contiguous():
                    raise _lib.NrgbdError("FusedAdam: parameter of shape %s is not contiguous" % (tuple(p.shape),))
                if not g.is_contiguous():
                    g = p.grad = g.contiguous()
                st = self._state_of(p)
                ptrs.append((p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), st["step"].data_ptr(), p.numel()))
                key.append(ptrs[-1])
            if not ptrs:
                continue
            key = tuple(key)
            hit = self._tables.get(gi)
            if hit is None or hit[0] != key:        # the host-side pointer arrays, rebuilt only when a tensor moved
                n = len(ptrs)
                arrs = [(ctypes.c_void_p * n)(*[t[k] for t in ptrs]) for k in range(5)]
                arrs.append((ctypes.c_long * n)(*[t[5] for t in ptrs]))
                hit = (key, arrs, n)
                self._tables[gi] = hit
            _, arrs, n = hit
            dev = group["params"][0].device
            with torch.cuda.device(dev):
                rc = lib.nrgbd_adam_step(arrs[0], arrs[1], arrs[2], arrs[3], arrs[4], arrs[5], n, float(group["lr"]),
                                         float(group["betas"][0]), float(group["betas"][1]), float(group["eps"]),
                                         float(group["weight_decay"]), int(bool(group["maximize"])),
                                         ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
            _lib.check(rc, "nrgbd_adam_step")
            for p in group["params"]:
                if p.grad is not None:
                    torch._C._increment_version(p)      # like torch.optim.Adam's in-place ops (see mark_updated)
        return loss
