"""Streaming depth filter: one object per video stream, DPV state resident on the GPU.

The reference's driver loop (test_KVNet.py:190-250) calls `test()` once per frame: ~600 kernel launches from
Python, one device->host probe (`valid_dpv`), a host-built point grid and its H2D copy.  `DepthStream` keeps the
same per-frame semantics (KVNET.forward + PREDICT, i.e. exactly `neuralrgbd_amd.test_step.test`) but

  * keeps BV_predict and the validity flag on its own side (no device->host read in the steady state),
  * captures the whole update-branch frame into ONE hipGraph (torch.cuda.CUDAGraph) after a warm-up frame, and
    replays it per frame: every kernel of libnrgbd_hip.so is capture-safe (no allocation, no host sync, no
    algorithm search — there is no vendor convolution on the path).  A capture that fails RAISES unless the stream
    was built with allow_eager_fallback=True (then it stays eager and `graph_error` says why).
  * reads the path's status word (BatchNorm variance collapse, nets.check_status) without stalling: every step
    queues a 4-byte copy to pinned host memory behind the frame and inspects the PREVIOUS step's copy.

  * optionally PIPELINES the two halves of consecutive frames of the one video (pipeline=True): the D-Net of frame t + 1
    (feature CNN, texel pack, fused warp + cost volume: KVNET.measure — nothing in it depends on the filter state) runs on
    a second HIP stream while the K-Net / DPV update / R-Net / PREDICT of frame t run on the first.
    Every frame is computed in full by the same kernels on the same inputs, so the outputs are the sequential ones bit for
    bit (tests/test_gpu_fullsize.py); step() then returns the result of the PREVIOUS call's frame (one frame of latency)
    and flush() the last one.  What it buys is the idle tail of every launch: +5 % frames/s at config B.

Config 5 of BASELINE.json (a 300-frame high-resolution stream) is this object in a loop.
"""
import math

import torch

from . import homography as warp_homo
from . import ops


def _capture_mode():
    """capture_error_mode of the hipGraph captures (see distributed.graph_capture_mode)."""
    from .distributed import graph_capture_mode
    return graph_capture_mode()


class DepthStream:
    def __init__(self, model, cam_intrinsics, d_candi, t_win_r=2, use_graph=True, device=None, copy_outputs=False,
                 allow_eager_fallback=False, pipeline=False):
        self.model = model
        self.cam = cam_intrinsics
        self.d_candi = d_candi
        self.t_win_r = t_win_r
        self.use_graph = use_graph
        # Once the hipGraph is active, step() hands out the graph's STATIC output buffers: the next step() overwrites them
        # in place, so a caller that keeps per-frame results (a list for export / evaluation) must either pass
        # copy_outputs=True (clones: +0.2 GB/s of HBM traffic at config B) or clone what it keeps.  The eager path
        # returns fresh tensors either way.
        self.copy_outputs = copy_outputs
        self.device = device if device is not None else next(model.parameters()).device
        self.bv_predict = None          # [1,D,h,w] log-DPV predicted for the next frame, or None (fresh stream)
        self._graph = None
        self._static = None
        self._eager_updates = 0
        self.graph_error = None
        self.allow_eager_fallback = allow_eager_fallback
        self._status_host = None        # pinned int32: the status word as of the previous step
        self._status_event = None
        # pipeline=True: two slots of static inputs + D-Net records, the slot of the frame whose back half is still owed, the
        # second HIP stream and the events that order the halves
        self.pipeline = bool(pipeline)
        self._slots = None
        self._pending = None            # slot index of the frame that has been measured but not filtered yet
        self._side = None
        self._ev_front = None
        self._ev_back = None

    def reset(self):
        """Invalid pose / new trajectory: drop the filter state (test_KVNet.py:241-246); a pending pipelined frame is dropped too."""
        self.bv_predict = None
        self._pending = None

    # ------------------------------------------------------------------ one frame, eager
    def _frame(self, ref, src, poses, pose_next, bv_predict):
        model = self.model
        with torch.no_grad():
            # test_KVNet.py:50 `.inverse()` as a path kernel (fixed operation order, capture-safe): inside the hipGraph
            pose_next_inv = ops.pose_inverse(pose_next)
            r_cur, r_kv, bv_cur, dpv = model(ref, src, poses, torch.zeros(1), cam_intrinsics=[self.cam],
                                             BV_predict=bv_predict, dpv_valid=bv_predict is not None)
            pad = math.log(1. / float(len(self.d_candi)))
            nxt = warp_homo.resample_vol_cuda(src_vol=dpv, rel_extM=pose_next_inv, cam_intrinsic=self.cam,
                                              d_candi=self.d_candi, padding_value=pad, clamp=(-1000., 0.)).unsqueeze(0)
        return r_kv, dpv, nxt

    def _capture(self, ref, src, poses, pose_next):
        st = {"ref": ref.clone(), "src": src.clone(), "poses": poses.clone(), "pose_next": pose_next.clone(),
              "bv": self.bv_predict.clone()}
        torch.cuda.synchronize(self.device)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode=_capture_mode()):
            st["out"] = self._frame(st["ref"], st["src"], st["poses"], st["pose_next"], st["bv"])
        st["consts"] = warp_homo.cache_snapshot()      # K / rays / d_candi the graph reads: kept alive with the graph
        self._graph, self._static = g, st

    # ------------------------------------------------------------------ pipelined halves (pipeline=True)
    def _front(self, sl):
        with torch.no_grad():
            sl["rec"] = self.model.measure(sl["ref"], sl["src"], sl["poses"])

    def _back(self, sl, bv):
        model = self.model
        with torch.no_grad():
            pose_next_inv = ops.pose_inverse(sl["pose_next"])
            r_cur, r_kv, bv_cur, dpv = model(sl["ref"], sl["src"], sl["poses"], torch.zeros(1), cam_intrinsics=[self.cam],
                                             BV_predict=bv, dpv_valid=True, measured=sl["rec"])
            pad = math.log(1. / float(len(self.d_candi)))
            nxt = warp_homo.resample_vol_cuda(src_vol=dpv, rel_extM=pose_next_inv, cam_intrinsic=self.cam,
                                              d_candi=self.d_candi, padding_value=pad, clamp=(-1000., 0.)).unsqueeze(0)
        sl["out"] = (r_kv, dpv, nxt)

    def _pipe_setup(self, ref, src, poses, pose_next):
        self._slots = [{"ref": torch.empty_like(ref), "src": torch.empty_like(src), "poses": torch.empty_like(poses),
                        "pose_next": torch.empty_like(pose_next), "gf": None, "gb": None} for _ in range(2)]
        self._bv = torch.empty_like(self.bv_predict)            # the filter state the back half reads (static for its graph)
        self._side = torch.cuda.Stream(self.device)
        self._ev_front = [torch.cuda.Event() for _ in range(2)]
        self._ev_back = torch.cuda.Event()
        self._ev_back.record(torch.cuda.current_stream(self.device))

    def _pipe_capture(self):
        """Both halves of both slots as hipGraphs (four captures; a slot's record lives in its front graph's pool and is read by its
        back graph: static buffers on both sides).  Needs every cache warm: called after eager pipelined frames."""
        torch.cuda.synchronize(self.device)
        for sl in self._slots:
            gf = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gf, capture_error_mode=_capture_mode()):
                self._front(sl)
            gb = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gb, capture_error_mode=_capture_mode()):
                self._back(sl, self._bv)
            sl["gf"], sl["gb"] = gf, gb
        self._consts = warp_homo.cache_snapshot()
        self._graph = True            # "the graphs are active" (reported by the callers through `_graph is not None`)

    def _step_pipelined(self, ref, src, poses, pose_next):
        """front(t) on the side stream, back(t - 1) on the caller's stream; returns the outputs of frame t - 1 (None on the first
        pipelined call, whose predecessor was handled by the sequential path)."""
        main = torch.cuda.current_stream(self.device)
        if self._slots is None:
            self._pipe_setup(ref, src, poses, pose_next)
        if (self.use_graph and self._graph is None and self.graph_error is None and self._eager_updates >= 2
                and self._pending is not None):
            try:
                # the pending frame's record was produced eagerly: re-measure it inside its slot after the capture (below)
                pend = self._pending
                self._pipe_capture()
                main.wait_event(self._ev_back)
                self._slots[pend]["gf"].replay()           # capture does not execute: the pending record is rebuilt in its static buffers
                self._ev_front[pend].record(main)
            except Exception as e:
                self.graph_error = repr(e)
                self._graph = None
                if not self.allow_eager_fallback:
                    raise
                print("[DepthStream] hipGraph capture failed, staying eager: %s" % self.graph_error)
        t = 0 if self._pending != 0 else 1                 # the free slot
        sl = self._slots[t]
        # frame t's inputs -> its slot (on the caller's stream, after the back half that last read this slot: same stream), then the
        # front half on the side stream
        sl["ref"].copy_(ref); sl["src"].copy_(src); sl["poses"].copy_(poses); sl["pose_next"].copy_(pose_next)
        if sl["gf"] is not None:
            ev_in = torch.cuda.Event()
            ev_in.record(main)
            with torch.cuda.stream(self._side):
                self._side.wait_event(ev_in)
                sl["gf"].replay()
                self._ev_front[t].record(self._side)
        else:
            # eager warm-up frames: both halves on the caller's stream, in order (tensors allocated under one stream and read under
            # another would need record_stream bookkeeping; only the graphs — static buffers — overlap)
            self._front(sl)
            self._ev_front[t].record(main)
        out = None
        if self._pending is not None:
            pl = self._slots[self._pending]
            main.wait_event(self._ev_front[self._pending])
            self._bv.copy_(self.bv_predict)
            if pl["gb"] is not None:
                pl["gb"].replay()
            else:
                self._back(pl, self._bv)
                self._eager_updates += 1
            r, dpv, nxt = pl["out"]
            self.bv_predict = nxt
            out = (r.clone(), dpv.clone()) if self.copy_outputs else (r, dpv)
        self._ev_back.record(main)
        self._pending = t
        self._probe_status()
        return out

    def flush(self):
        """pipeline=True: run the back half of the frame that is still owed and return its outputs (None if nothing is pending)."""
        if not self.pipeline or self._pending is None:
            return None
        main = torch.cuda.current_stream(self.device)
        pl = self._slots[self._pending]
        main.wait_event(self._ev_front[self._pending])
        self._bv.copy_(self.bv_predict)
        if pl["gb"] is not None:
            pl["gb"].replay()
        else:
            self._back(pl, self._bv)
        r, dpv, nxt = pl["out"]
        self.bv_predict = nxt.clone()
        self._pending = None
        self._ev_back.record(main)
        return r.clone(), dpv.clone()

    def _probe_status(self):
        """Deferred, stall-free read of the status word: raise on what the PREVIOUS step's copy shows, queue this step's copy."""
        from . import nets
        if self._status_event is not None and self._status_event.query() and int(self._status_host[0]) != 0:
            nets.check_status(self.device)          # reads, clears and raises
        word = nets.status_word(self.device)
        if self._status_host is None:
            self._status_host = torch.zeros(1, dtype=torch.int32).pin_memory()
            self._status_event = torch.cuda.Event()
        self._status_stage = word.clone()           # a device-side copy first (see nets.check_status: no memcpy from the word itself)
        self._status_host.copy_(self._status_stage, non_blocking=True)
        self._status_event.record(torch.cuda.current_stream(self.device))

    def check(self):
        """Synchronous form of the status probe (end of a sequence): raises NrgbdError on a reported variance collapse."""
        from . import nets
        nets.check_status(self.device)

    # ------------------------------------------------------------------ public step
    def step(self, ref_frame, src_frames, src_cam_poses, cam_pose_next=None):
        """ref_frame [1,3,H,W], src_frames [1,V,3,H,W], src_cam_poses [1,V,4,4] (device tensors).
        Returns (refined DPV [1,D,H,W], DPV [1,D,h,w]); the predicted state for the next frame is kept inside.
        pipeline=True: the pair belongs to the PREVIOUS call's frame (None when there is none yet); flush() returns the last one.
        With the hipGraph active and copy_outputs=False the returned tensors are only valid until the next step()."""
        pose = src_cam_poses[0, self.t_win_r] if cam_pose_next is None else cam_pose_next
        # the pose of the next reference frame, NOT inverted: the inversion happens inside the frame (nrgbd_pose_inverse); a host
        # tensor is accepted as before
        pose_next = pose.to(device=self.device, dtype=torch.float32).contiguous()
        if self.bv_predict is None:                      # first window of the stream: D-Net only
            r, dpv, nxt = self._frame(ref_frame, src_frames, src_cam_poses, pose_next, None)
            self.bv_predict = nxt
            self._probe_status()
            return r, dpv
        if self.pipeline:
            return self._step_pipelined(ref_frame, src_frames, src_cam_poses, pose_next)
        if self.use_graph and self._graph is None and self.graph_error is None and self._eager_updates >= 1:
            try:
                self._capture(ref_frame, src_frames, src_cam_poses, pose_next)
            except Exception as e:
                self.graph_error = repr(e)
                self._graph = None
                if not self.allow_eager_fallback:      # a library user must not silently run at eager speed
                    raise
                print("[DepthStream] hipGraph capture failed, staying eager: %s" % self.graph_error)
        if self._graph is not None:
            st = self._static
            st["ref"].copy_(ref_frame); st["src"].copy_(src_frames); st["poses"].copy_(src_cam_poses)
            st["pose_next"].copy_(pose_next); st["bv"].copy_(self.bv_predict)
            self._graph.replay()
            r, dpv, nxt = st["out"]
            self.bv_predict = nxt        # static output buffer: copied into st["bv"] at the next step
            self._probe_status()
            if self.copy_outputs:
                return r.clone(), dpv.clone()
            return r, dpv                # valid until the next step() (see copy_outputs)
        r, dpv, nxt = self._frame(ref_frame, src_frames, src_cam_poses, pose_next, self.bv_predict)
        self._eager_updates += 1
        self.bv_predict = nxt
        self._probe_status()
        return r, dpv
