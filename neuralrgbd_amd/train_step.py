"""One training iteration of KVNET — mirror of code/train_utils/train_KVNet.py:20-203 (`train`) for the
one-process-per-GPU design.

Differences from the reference, all forced by replacing single-process `DataParallel` (train_KVNet.py:261-262):
  * each process owns ONE trajectory (N = 1: the reference asserts N = 1 per replica anyway, KVNET.py:116);
  * the `loss / nGPU` + DataParallel reduce-add (:149-152) becomes an all-reduce of the gradient (sum / world)
    right after `backward()` — pass `grad_reducer=neuralrgbd_amd.distributed.GradAllReduce(model)`;
  * only loss_type 'NLL' (what every training script of the reference uses; 'L1' needs the DGF net).
Forward/backward run the fused HIP sampling kernels behind autograd (neuralrgbd_amd.autograd); the PREDICT step
at the end is the same single launch as at inference.
"""
import math

import torch
import torch.nn.functional as F

from . import homography as warp_homo
from . import ops
from .autograd import nll_loss_d, pack_cache
from .misc import depth_val_regression, valid_dpv


def _capture_mode():
    """capture_error_mode of the hipGraph captures (see distributed.graph_capture_mode)."""
    from .distributed import graph_capture_mode
    return graph_capture_mode()


def _nll_terms(d_dpv, dmap_cur_refined, kv_dpv, dmap_refined, depth_ref, depth_ref_imgsize, valid):
    """train_KVNet.py:103-120: NLL on the 1/4-res DPV and on its R-Net refinement, for the measurement and (update branch)
    for the filtered volume."""
    loss = nll_loss_d(d_dpv, depth_ref, ignore_index=0)
    loss = loss + nll_loss_d(dmap_cur_refined, depth_ref_imgsize, ignore_index=0)
    if valid:
        loss = loss + nll_loss_d(kv_dpv, depth_ref, ignore_index=0)
        loss = loss + nll_loss_d(dmap_refined, depth_ref_imgsize, ignore_index=0)
    return loss


def _predict(kv, pose_next, cam, d_candi):
    """PREDICT on the detached DPV (train_KVNet.py:155-171): fixed-order pose inverse + one resampling launch."""
    rel_Rt = ops.pose_inverse(pose_next.to(dtype=torch.float32).contiguous())
    return warp_homo.resample_vol_cuda(src_vol=kv, rel_extM=rel_Rt, cam_intrinsic=cam, d_candi=d_candi,
                                       padding_value=math.log(1. / float(len(d_candi))), clamp=(-1000., 0.)).unsqueeze(0)


def _train_windows(A, dev, model_KV, t_win_r, d_candi, Ref_Dats, Src_Dats, poses_all, BVs_predict, Cam_Intrinsics):
    """Forward, losses and backward of the A windows of one optimizer step (train() below); gradients accumulate in .grad."""
    outs = []
    for b in range(A):
        ref_frame = Ref_Dats[b]['img'].to(dev)
        src_frames = torch.cat(tuple(f['img'] for f in Src_Dats[b]), dim=0).unsqueeze(0).to(dev)
        poses = poses_all[b:b + 1]
        bv = BVs_predict[b] if isinstance(BVs_predict, (list, tuple)) else (
            BVs_predict[b:b + 1] if isinstance(BVs_predict, torch.Tensor) and A > 1 else BVs_predict)
        if isinstance(bv, torch.Tensor) and bv.dim() == 3:
            bv = bv.unsqueeze(0)
        cam = Cam_Intrinsics[b] if len(Cam_Intrinsics) == A else Cam_Intrinsics[0]
        valid = valid_dpv(bv) if isinstance(bv, torch.Tensor) else False
        dmap_cur_refined, dmap_refined, d_dpv, kv_dpv = model_KV(
            ref_frame=ref_frame, src_frames=src_frames, src_cam_poses=poses, BatchIdx=torch.zeros(1),
            cam_intrinsics=[cam], BV_predict=bv if valid else None, dpv_valid=True if valid else None)
        depth_ref = Ref_Dats[b]['dmap'].to(dev)                        # [1,h,w] int64 bin indices, 0 = ignore
        depth_ref_imgsize = Ref_Dats[b]['dmap_imgsize_digit'].to(dev)  # [1,H,W]
        loss = _nll_terms(d_dpv, dmap_cur_refined, kv_dpv, dmap_refined, depth_ref, depth_ref_imgsize, valid)
        loss.backward()                # accumulates: the mean over the A windows is taken once, after the all-reduce
        with torch.no_grad():
            kv = kv_dpv.detach()
            outs.append((dmap_cur_refined.detach(), _predict(kv, poses[0, t_win_r], cam, d_candi), loss.detach(),
                         depth_val_regression(kv, d_candi, BV_log=True),
                         depth_val_regression(dmap_refined.detach(), d_candi, BV_log=True)))
    return outs


def train(nGPU, model_KV, optimizer_KV, t_win_r, d_candi, Ref_Dats, Src_Dats, Src_CamPoses, BVs_predict,
          Cam_Intrinsics, refine_dup=False, weight_var=.001, loss_type='NLL', mGPU=False,
          Cam_Intrinsics_spatial_up=None, return_confmap_up=False, grad_reducer=None, accum_steps=1):
    """Returns (r_dpv, BVs_predict_out, loss, dmap_kv_lowres, dmap_kv_highres) — the two depth maps are device
    tensors (the reference stacks them with the ground truth into numpy arrays for TensorBoard).

    accum_steps = A > 1 (BASELINE config 4: global batch 32 = 8 GPUs x 4): the call takes A trajectories the way the
    reference's batch dimension does — Ref_Dats / Src_Dats lists of length A, Src_CamPoses [A,V,4,4], BVs_predict None, a
    tensor [A,D,h,w] or a list of A (tensor | None) — and runs them as A SEQUENTIAL N = 1 windows (KVNET asserts N = 1 per
    forward), each with its own BV_predict: A forward/backward passes accumulate into the same gradient, ONE all-reduce,
    division by A x world, ONE optimizer step.  Outputs are concatenated along the batch dimension; `loss` is the mean."""
    if loss_type != 'NLL' or refine_dup:
        raise NotImplementedError("only the NLL loss without depth up-sampling is on this path")
    A = int(accum_steps)
    if A < 1:
        raise ValueError("accum_steps must be >= 1")
    if len(Ref_Dats) != A or len(Src_Dats) != A:
        raise AssertionError("one trajectory per accumulation step and process (N = 1 per forward): %d windows for accum_steps=%d"
                             % (len(Ref_Dats), A))
    dev = next(model_KV.parameters()).device
    poses_all = Src_CamPoses.to(dev)
    if grad_reducer is not None and hasattr(grad_reducer, "prepare"):
        # .grad = zeroed views into the all-reduce buckets (no per-step flatten / scatter)
        grad_reducer.prepare(A) if A > 1 else grad_reducer.prepare()
    else:
        optimizer_KV.zero_grad()

    with pack_cache():                # the A windows run on the same weights: their packed streams are built by the first one
        outs = _train_windows(A, dev, model_KV, t_win_r, d_candi, Ref_Dats, Src_Dats, poses_all, BVs_predict, Cam_Intrinsics)

    if grad_reducer is not None:
        grad_reducer()                 # RCCL all-reduce (sum / (A * world)) of the 21 MB fp32 gradient: buckets whose gradients
                                       # were complete started from the last window's backward hooks; this waits for all of them
    if A > 1 and not (grad_reducer is not None and hasattr(grad_reducer, "prepare")):
        # a plain callable reducer (sum / world) knows nothing about the accumulation: the mean over the A windows is taken here
        torch._foreach_div_([p.grad for p in model_KV.parameters() if p.grad is not None], float(A))
    optimizer_KV.step()
    if A == 1:
        return outs[0]
    r_dpv, pred, loss, lo, hi = zip(*outs)
    return torch.cat(r_dpv, 0), torch.cat(pred, 0), torch.stack(loss).mean(), torch.cat(lo, 0), torch.cat(hi, 0)


class TrainGraph:
    """The update-branch training iteration (forward under autograd, 4 NLL terms, backward, Adam, PREDICT) captured once
    into a hipGraph and replayed: one training step launches ~1,300 kernels, and launched from Python the host, not the
    GPU, sets the pace (tools/bench_train.py: 62 ms eager vs 58 ms of kernels).

    Two forms.
      * ONE graph (grad_reducer None, accum_steps 1): forward + backward + Adam + PREDICT, gradients in the graph's pool.
      * SPLIT (a `grad_reducer` for N > 1 ranks and / or accum_steps = A > 1; BASELINE config 4 = 8 GPUs x A = 4): graph 1 =
        forward + losses + backward + PREDICT of ONE window, accumulating into persistent gradient buffers (the reducer's
        bucket views, so the message IS the gradient); it is replayed A times, once per window; then the bucketed all-reduce
        (RCCL; eager, between the graphs — replays run no autograd hooks, so every bucket is launched by `reducer()` in index
        order; 21 MB over xGMI is ~0.3 ms of a ~150 ms step, there is nothing worth overlapping) and the division by
        A x world; graph 2 = the optimizer step.  `step_windows()` drives this form.
    The optimizer is neuralrgbd_amd.optim.FusedAdam (one multi-tensor kernel, capturable as it is) or a torch optimizer built
    with `capturable=True`.  Inputs are copied into static buffers; the pose inverse for PREDICT
    is computed outside the graph.  Outputs are static tensors that the next replay overwrites.

    Capture needs state that only an executed iteration creates: Adam's exp_avg / exp_avg_sq / step (created inside a
    capture they would become graph nodes that reset the moments at every replay), the packed-weight and interpolation
    caches (their first use uploads from the host) and the vendor convolutions' algorithm choice.  Therefore the first
    `warmup` calls of step() (default 1) are ordinary EAGER training iterations on their own windows — real steps, not
    duplicates — and the graph is captured at the first call after them, when every parameter has optimizer state
    (checked; a RuntimeError names the parameter otherwise).
    """

    def __init__(self, model, optimizer, t_win_r, d_candi, cam_intrinsics, warmup=1, grad_reducer=None, accum_steps=1):
        self.model, self.opt, self.t_win_r, self.d_candi, self.cam = model, optimizer, t_win_r, d_candi, cam_intrinsics
        self._graph = None
        self._st = None
        self._warmup = max(0, int(warmup))
        self._eager_steps = 0
        self.reducer = grad_reducer
        self.accum = int(accum_steps)
        if self.accum < 1:
            raise ValueError("accum_steps must be >= 1")
        self.split = grad_reducer is not None or self.accum > 1
        self._g_opt = None
        self._g_rest = None
        self._grads = None

    def _optimizer_ready(self):
        """Every trainable parameter has populated optimizer state (else capture would record its creation)."""
        names = {p: n for n, p in self.model.named_parameters()}
        if hasattr(self.opt, "init_state"):     # neuralrgbd_amd.optim.FusedAdam: the state can simply be created now
            self.opt.init_state()
        for group in self.opt.param_groups:
            for p in group["params"]:
                if p.requires_grad and not self.opt.state.get(p):
                    return names.get(p, "<unnamed %s>" % (tuple(p.shape),))
        return None

    def _fwd_bwd(self, st):
        """forward + 4 NLL terms + backward + PREDICT of one window (no optimizer step)."""
        model = self.model
        r_cur, r_kv, d_dpv, kv_dpv = model(ref_frame=st["ref"], src_frames=st["src"], src_cam_poses=st["poses"],
                                           BatchIdx=torch.zeros(1), cam_intrinsics=[self.cam], BV_predict=st["bv"],
                                           dpv_valid=True)
        loss = _nll_terms(d_dpv, r_cur, kv_dpv, r_kv, st["dmap"], st["dmap_full"], True)
        loss.backward()
        with torch.no_grad():
            nxt = warp_homo.resample_vol_cuda(src_vol=kv_dpv.detach(), rel_extM=st["inv"], cam_intrinsic=self.cam,
                                              d_candi=self.d_candi, padding_value=math.log(1. / float(len(self.d_candi))),
                                              clamp=(-1000., 0.)).unsqueeze(0)
        return loss.detach(), nxt

    def _iteration(self, st):
        out = self._fwd_bwd(st)
        self.opt.step()
        return out

    def _mark_updated(self):
        """A replay of the captured optimizer step runs no Python: the parameters' version counters (the keys of the inference
        caches in nets.py) are advanced here."""
        if hasattr(self.opt, "mark_updated"):
            self.opt.mark_updated()
        else:
            for group in self.opt.param_groups:
                for p in group["params"]:
                    if p.requires_grad:
                        torch._C._increment_version(p)

    def _upload_constants(self, dev):
        # per-trajectory constants (intrinsics, ray table, d_candi) are uploaded once and cached per dict: do it now,
        # an upload inside the capture is not allowed
        for cam in (self.cam, getattr(self.model.d_net, "cam_intrinsics", self.cam)):
            warp_homo._cam_dev(cam, dev)
        for dc in (self.d_candi, getattr(self.model.d_net, "d_candi", self.d_candi), getattr(self.model, "d_candi", self.d_candi)):
            warp_homo._d_candi_dev(dc, dev)

    def _needs_eager(self):
        if self._graph is not None:
            return False
        if self._eager_steps < self._warmup or self._optimizer_ready() is not None:
            if self._eager_steps >= max(self._warmup, 2):
                raise RuntimeError("TrainGraph: parameter %s still has no optimizer state after %d eager iterations "
                                   "(unused in the loss?): the iteration cannot be captured" %
                                   (self._optimizer_ready(), self._eager_steps))
            return True
        return False

    def step(self, ref_frame, src_frames, poses, dmap, dmap_full, bv_predict):
        if self.split:
            if self.accum != 1:
                raise ValueError("accum_steps = %d: pass the windows of a step together (step_windows)" % self.accum)
            loss, nxt = self.step_windows([(ref_frame, src_frames, poses, dmap, dmap_full, bv_predict)])
            return loss, nxt[0]
        inv = ops.pose_inverse(poses[0, self.t_win_r].to(dtype=torch.float32).contiguous())
        if self._needs_eager():
            self.opt.zero_grad(set_to_none=True)
            out = self._iteration({"ref": ref_frame, "src": src_frames, "poses": poses, "dmap": dmap,
                                   "dmap_full": dmap_full, "bv": bv_predict, "inv": inv})
            self._eager_steps += 1
            return out
        if self._graph is None:
            st = {"ref": ref_frame.clone(), "src": src_frames.clone(), "poses": poses.clone(), "dmap": dmap.clone(),
                  "dmap_full": dmap_full.clone(), "bv": bv_predict.clone(), "inv": inv.clone()}
            self._upload_constants(ref_frame.device)
            self.opt.zero_grad(set_to_none=True)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode=_capture_mode()):
                st["out"] = self._iteration(st)      # gradients are allocated in the graph's pool and rewritten per replay
            st["consts"] = warp_homo.cache_snapshot()   # K / rays / d_candi the graph reads: kept alive with the graph
            self._graph, self._st = g, st
            # capture does not execute: fall through to the first replay with the same inputs
        st = self._st
        st["ref"].copy_(ref_frame); st["src"].copy_(src_frames); st["poses"].copy_(poses)
        st["dmap"].copy_(dmap); st["dmap_full"].copy_(dmap_full); st["bv"].copy_(bv_predict); st["inv"].copy_(inv)
        self._graph.replay()
        self._mark_updated()
        return st["out"]

    # ------------------------------------------------------------------ split form: N > 1 ranks and / or accumulation
    def _zero_grads(self):
        """Gradients of the split form live OUTSIDE the graphs: graph 1 accumulates into them over its A replays, the
        all-reduce runs on them, graph 2 reads them."""
        if self.reducer is not None:
            # replays run no hooks; nothing may be launched from capture-time hooks.  Scoped to this step: step_windows restores
            # the caller's setting when it returns (an eager train(..., grad_reducer=reducer) afterwards overlaps again)
            self.reducer.hold = True
            self.reducer.prepare(self.accum)
            return
        if self._grads is None:
            self._grads = [(p, torch.zeros_like(p)) for p in {id(q): q for q in self.model.parameters() if q.requires_grad}.values()]
        torch._foreach_zero_([g for _, g in self._grads])
        for p, g in self._grads:
            p.grad = g

    def _reduce_grads(self):
        if self.reducer is not None:
            self.reducer()                            # every bucket, in index order; / (A x world)
        elif self.accum > 1:
            torch._foreach_div_([g for _, g in self._grads], float(self.accum))

    def step_windows(self, windows):
        """One optimizer step over `accum_steps` windows [(ref, src, poses, dmap, dmap_full, bv_predict), ...] of this rank.
        Returns (mean loss, [BV_predict of each window's next frame]) — clones, valid until overwritten by the caller."""
        if len(windows) != self.accum:
            raise ValueError("%d windows for accum_steps=%d" % (len(windows), self.accum))
        hold0 = getattr(self.reducer, "hold", None)
        try:
            return self._step_windows(windows)
        finally:
            if self.reducer is not None and hold0 is not None:
                self.reducer.hold = hold0

    def _step_windows(self, windows):
        invs = [ops.pose_inverse(w[2][0, self.t_win_r].to(dtype=torch.float32).contiguous()) for w in windows]
        keys = ("ref", "src", "poses", "dmap", "dmap_full", "bv")
        if self._needs_eager():
            self._zero_grads()
            with pack_cache():
                outs = [self._fwd_bwd(dict(zip(keys, w), inv=inv)) for w, inv in zip(windows, invs)]
            self._reduce_grads()
            self.opt.step()
            self._eager_steps += 1
            return torch.stack([o[0] for o in outs]).mean(), [o[1] for o in outs]
        if self._graph is None:
            w0 = windows[0]
            st = dict(zip(keys, (t.clone() for t in w0)), inv=invs[0].clone())
            self._upload_constants(w0[0].device)
            self._zero_grads()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            packed = {}                               # the weight streams the first window's graph writes (autograd.pack_cache)
            with torch.cuda.graph(g, capture_error_mode=_capture_mode()), pack_cache(packed):
                st["out"] = self._fwd_bwd(st)         # accumulates into the persistent gradients
            g_rest = None
            if self.accum > 1 and packed:
                # windows 2 .. A of a step run on the same weights: a second capture of the same iteration with the cache filled holds
                # no packing launch and reads the streams the first graph wrote (same pool; replay order = capture order)
                g_rest = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g_rest, pool=g.pool(), capture_error_mode=_capture_mode()), pack_cache(packed):
                    st["out_rest"] = self._fwd_bwd(st)
            g2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g2, pool=g.pool(), capture_error_mode=_capture_mode()):
                self.opt.step()
            st["consts"] = warp_homo.cache_snapshot()
            st["packed"] = packed
            self._graph, self._g_rest, self._g_opt, self._st = g, g_rest, g2, st
        st = self._st
        self._zero_grads()
        losses, preds = [], []
        for i, (w, inv) in enumerate(zip(windows, invs)):
            for k, t in zip(keys, w):
                st[k].copy_(t)
            st["inv"].copy_(inv)
            if i == 0 or self._g_rest is None:
                self._graph.replay(); out = st["out"]
            else:
                self._g_rest.replay(); out = st["out_rest"]
            losses.append(out[0].clone()); preds.append(out[1].clone())
        self._reduce_grads()
        self._g_opt.replay()
        self._mark_updated()
        return torch.stack(losses).mean(), preds
