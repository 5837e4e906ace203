#!/usr/bin/env python
"""bench.py — depth frames/s of the plane-sweep path on MI355X + roofline of the fused warp kernel.

    python bench.py --gpus N --steps K --warmup W            (N=1)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One *step* = one depth frame of one video stream = one call of the reference's
test_utils/test_KVNet.py::test in the update branch: KVNET.forward with a valid BV_predict (D-Net,
K-Net, DPV update, both R-Net calls) + the PREDICT resample.  Inputs are synthetic windows that are
resident in HBM before the timed region; weights are seeded random (no checkpoints offline).
Each rank streams its own independent video (replicas, no data-path collective) => weak scaling,
value = all ranks' frames / max-over-ranks time.

Workload (BASELINE.json configs[1]): plane-sweep grid 256x192 (image
1024x768), 64 depth candidates, 5-frame window (1 reference + 4 sources), fp32.

The JSON line carries
  roofline      HBM roofline of the dominant sampling kernel (fused warp + cost volume + log-softmax):
                algorithmic bytes 4[(V+1)*67*hw + D*hw] per launch / its mean launch duration, measured
                with HIP events on the launch stream inside the timed steps (peak 8.0 TB/s);
  cpu_baseline  the CPU oracle (oracle/kvnet_oracle.py: the reference algorithm restated on torch-CPU
                + the C sampling oracle) timed on this node's host cores on ONE update frame of the
                same workload (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    "S": dict(H=256, W=384, D=64, d_min=0.1, d_max=5.0, name="ScanNet demo 384x256 image, grid 96x64x64"),
    "B": dict(H=768, W=1024, D=64, d_min=0.1, d_max=5.0, name="plane-sweep grid 256x192x64cand (image 1024x768)"),
    "K": dict(H=256, W=768, D=64, d_min=1.0, d_max=60.0, name="KITTI 768x256 image, grid 192x64x64"),
    "H": dict(H=480, W=640, D=128, d_min=0.1, d_max=5.0, name="480x640 image, grid 160x120x128"),
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense fp32 matrix peak (256 CUs x 256 FLOP/clk x 2.4 GHz)
# HBM bytes per launch of the fused warp + cost-volume kernel at config B from the L2's fabric-side counters
# (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes: tools/pmc_costvol.sh, summary in
# profiles/r1_pmc_summary.txt, round-1 final kernel): FETCH_SIZE 665,420 KB, doubled as MI355X_MICROARCH.md §HBM prescribes for gfx950
# (calibrated in this access pattern on logsoftmax_d: 6,283 KB reported for the 12,583 KB it reads), WRITE_SIZE
# 12,288 KB (= the cost volume exactly).  Counters cannot be read from inside this process, hence a constant.
PMC_TRAFFIC_BYTES = {"B": 2 * 665420 * 1024 + 12288 * 1024}


def costvol_bytes(V, C, D, h, w):
    """Algorithmic bytes of the fused warp + cost-volume kernel."""
    return 4 * ((V + 1) * C * h * w + D * h * w)


class KernelTimer:
    """Duration of one kernel of the frame (ops.costvol: the sampling kernel the metric names; ops.conv3d: the K-Net
    layer that dominates the frame time) from HIP events on torch's current stream (= the
    stream the kernel is launched on).  Frames are replayed as one hipGraph, inside which single kernels cannot be
    bracketed, so the wrapper remembers the arguments of the last in-model launch (same tensors, same shapes) and
    `measure()` re-launches exactly that K times back to back between two events right after the timed region."""

    def __init__(self, keep=lambda a, k: True):
        self.last = None
        self.fn = None
        self.keep = keep

    def wrap(self, fn):
        self.fn = fn

        def remembering(*a, **k):
            if self.keep(a, k):
                self.last = (a, k)
            return fn(*a, **k)
        return remembering

    def measure(self, launches):
        a, k = self.last
        self.fn(*a, **k)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(launches):
            self.fn(*a, **k)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / launches


def cpu_baseline(cfg, cam, d_candi, sd, window, bv_pred, sigma):
    """Update-branch frames of the SAME workload through the CPU oracle on the host cores: one warm-up frame, then the
    median (= mean) of two timed frames.  Returns (json dict, the oracle's outputs of the frame) — the
    outputs are what the `parity` block of the JSON line is computed against."""
    from oracle import cpu_oracle, kvnet_oracle
    # torch-CPU convolutions stop scaling long before this host's 256 hardware threads: measured on the
    # MI355X node (2 x EPYC 9575F) with tools/cpu_threads_probe.py, one config-S frame takes 1.49 / 1.47 /
    # 2.33 / 4.6 s at 16 / 32 / 64 / 128 threads and 8x longer at 256 — so the baseline uses its best: 32.
    # "port": the oracle is a restatement (C sampling + the ATen CPU ops the reference itself calls); the unmodified
    # reference cannot travel to the GPU box (/root/reference does not exist there).
    cores = min(32, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    cpu_oracle.set_threads(cores)
    sd_cpu = {k: v.cpu() for k, v in sd.items()}
    r, s, p = (t.cpu() for t in window)
    bv = bv_pred.cpu()
    times, out = [], None
    for _ in range(3):
        t0 = time.time()
        out = kvnet_oracle.step(sd_cpu, r, s, p, cam, d_candi, sigma, bv)
        times.append(time.time() - t0)
    dt = 0.5 * (times[1] + times[2])
    return {"value": 1.0 / dt, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": "update-branch frames (KVNET.forward + PREDICT) of config %s through oracle/kvnet_oracle.py: 1 warm-up "
                      "(%.1f s) + 2 timed (%.1f, %.1f s), median reported" % (cfg, times[0], times[1], times[2])}, out


def parity_block(cfg, gpu, oracle_out):
    """GPU frame vs the oracle's frame on the same window and the same filter state: max / mean |d| and arg-max
    depth-index mismatches of BV_cur, DPV, BV_predict and the refined DPV (BASELINE.json gates: L1 < 1e-4, arg-max exact)."""
    names = ("refined", "dpv", "bv_cur", "bv_predict")
    blk = {"config": cfg, "against": "oracle/kvnet_oracle.py (CPU), same window, same BV_predict"}
    for name, g, o in zip(names, gpu, oracle_out):
        g, o = g[0].float().cpu(), o[0].float()
        d = (g - o).abs()
        blk[name] = {"max": float(d.max()), "mean": float(d.mean()),
                     "argmax_mismatch": int((g.argmax(0) != o.argmax(0)).sum()), "pixels": int(g[0].numel())}
    # gates: L1 < 1e-4 on every volume; arg-max identical on the depth volumes (BV_predict's faces are overwritten with a
    # constant, its arg-max is a tie by construction and is reported only)
    blk["pass"] = all(blk[n]["mean"] < 1e-4 for n in names) and \
        all(blk[n]["argmax_mismatch"] == 0 for n in ("refined", "dpv", "bv_cur"))
    return blk


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="B", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel from Python instead of replaying the captured hipGraph")
    ap.add_argument("--streams", type=int, default=1, help="independent video streams per GPU, each on its own HIP stream "
                    "(a step is then one frame of EVERY stream; the extra streams fill the tails of each other's kernels)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    import neuralrgbd_amd
    from neuralrgbd_amd import camera, ops, synth
    from neuralrgbd_amd.streaming import DepthStream

    cfg = CONFIGS[args.config]
    H, W, D, V = cfg["H"], cfg["W"], cfg["D"], 4
    h, w = H // 4, W // 4
    sigma = 10.0
    cam = camera.scannet_intrinsics(w, h) if args.config != "K" else camera.kitti_intrinsics(w, h)
    d_candi = np.linspace(cfg["d_min"], cfg["d_max"], D)
    model = neuralrgbd_amd.KVNET(64, cam, d_candi, sigma, 64, None, if_refined=True, refineNet_name="DPV", t_win_r=2)
    sd = synth.seeded_state_dict(model, 0)
    model.load_state_dict(sd)
    model = model.to(dev)  # stays in train() mode like the reference

    # a short ring of synthetic windows per rank, resident in HBM (different video per rank)
    ring = [tuple(t.to(dev) for t in synth.noise_window(1000 * rank + i, H, W, V)) for i in range(2)]

    timer = KernelTimer()
    ops.costvol = timer.wrap(ops.costvol)
    # the K-Net's plain 64->64 layer (BatchNorm+ReLU prologue, no residual operand): 6 of its 12 layers
    knet_timer = KernelTimer(keep=lambda a, k: a[0].shape[-1] == 64 and k.get("res") is None and k.get("x_ss") is not None)
    ops.conv3d = knet_timer.wrap(ops.conv3d)

    # the streaming driver: same per-frame work as test_utils/test_KVNet.py::test (R_net=True), state resident,
    # the update-branch frame captured into one hipGraph after an eager warm-up frame.  Extra streams per GPU are
    # further independent videos with their own model replica, filter state, graph and HIP stream.
    import copy
    S = max(1, args.streams)
    models = [model] + [copy.deepcopy(model) for _ in range(S - 1)]
    hip_streams = [torch.cuda.current_stream(dev)] + [torch.cuda.Stream(dev) for _ in range(S - 1)]
    streams = [DepthStream(m, cam, d_candi, t_win_r=2, use_graph=not args.no_graph, device=dev) for m in models]
    stream = streams[0]

    def frame(i):
        out = None
        for k in range(S):
            r, s_, p = ring[(i + k) % len(ring)]
            with torch.cuda.stream(hip_streams[k]):
                out = streams[k].step(r, s_, p)
        return out

    frame(0)                       # first window of the stream: D-Net only, creates the filter state
    for i in range(max(args.warmup, 2)):   # >= 2: one eager update frame, then the capture frame
        frame(i + 1)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        frame(i)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    pred = stream.bv_predict
    assert bool(torch.isfinite(pred).all()), "filter state went non-finite"

    if rank == 0:
        n_k = max(args.steps, 5)
        k_ms = timer.measure(n_k)
        algo = costvol_bytes(V, 67, D, h, w)
        achieved = algo / (k_ms * 1e-3) / 1e9
        line = {
            "metric": "depth frames/sec @256x192x64cand, 5-view window; warp-kernel HBM GB/s vs peak",
            "value": args.steps * S * world / dt, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": cfg["name"], "config_id": args.config, "grid_hw": [h, w], "depth_candidates": D,
                       "views": V + 1, "streams_per_gpu": S, "launch": "hipGraph replay" if stream._graph is not None else "eager",
                       "parallelism": "replicas x%d (independent video streams)" % world,
                       "peak_hbm_gb": torch.cuda.max_memory_allocated(dev) / 1e9},
            "roofline": {"bound": "hbm", "kernel": "costvol_lds<17,L2> + logsoftmax_d (fused warp + cost volume, log-softmax)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "algorithmic_bytes": algo, "kernel_ms": k_ms, "launches_timed": n_k,
                         "timing": "HIP events around back-to-back launches of the frame's own costvol call (log-softmax launch included), right after the timed region",
                         "traffic": PMC_TRAFFIC_BYTES.get(args.config),
                         "traffic_source": "rocprofv3 PMC (FETCH_SIZE x2 + WRITE_SIZE), profiles/r1_pmc_summary.txt"
                         if args.config in PMC_TRAFFIC_BYTES else None},
        }
        if knet_timer.last is not None:   # secondary roofline: the matrix-core kernel that takes most of the frame
            c_ms = knet_timer.measure(5)
            flops = 2.0 * D * h * w * 64 * 64 * 27
            tf = flops / (c_ms * 1e-3) / 1e12
            line["roofline_mfma"] = {"bound": "mfma", "kernel": "conv3d_mfma_kernel<64> (one K-Net 3x3x3 64->64 layer; the 10 such "
                                     "layers are ~2/3 of the frame)", "achieved": tf, "peak": MFMA_F32_PEAK_TFLOPS,
                                     "unit": "TFLOP/s", "frac": tf / MFMA_F32_PEAK_TFLOPS, "flops": flops, "kernel_ms": c_ms,
                                     "launches_timed": 5, "timing": "HIP events around back-to-back re-launches of the frame's own layer call"}
        if world == 1 and not args.no_cpu_baseline:
            # the same frame on both sides: window ring[0] filtered with the stream's current state
            pred = pred.clone()
            r_, s_, p_ = ring[0]
            with torch.no_grad():
                _, r_kv, bv_cur, dpv = model(r_, s_, p_, torch.zeros(1), cam_intrinsics=[cam], BV_predict=pred, dpv_valid=True)
                from neuralrgbd_amd import homography as warp_homo
                nxt = warp_homo.resample_vol_cuda(dpv, torch.linalg.inv(p_[0, 2]), cam_intrinsic=cam, d_candi=d_candi,
                                                  padding_value=float(np.log(1.0 / D)), clamp=(-1000., 0.)).unsqueeze(0)
            torch.cuda.synchronize()
            line["cpu_baseline"], o = cpu_baseline(args.config, cam, d_candi, sd, ring[0], pred, sigma)
            line["parity"] = parity_block(args.config, (r_kv, dpv, bv_cur, nxt), o)
        print(json.dumps(line))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
