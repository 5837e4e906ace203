#!/usr/bin/env python
"""Why is the K-Net layer kernel slower inside the frame than alone?  (2.50 vs 2.17 ms.)

Times ONE launch of conv_wino_dw (plain 64->64 layer at config B) under different histories and reads the shader clock right
behind it (tools/probes/libclock_probe.so: shader cycles vs the 100 MHz wall counter over ~20 us on every CU):
  burst      5 launches after 2 s of idle              (what tools/bench_wino.py measures)
  sustained  600 launches back to back (~1.4 s)        per-launch HIP events: how the time moves as the part heats / hits its cap
  gapped     100 launches with 20 ms of idle between    the same kernel on a cool part
  frame      30 frames of the config-B stream, then 5 launches of the frame's own layer call (what bench.py's roofline_mfma does)
  zeros      sustained, all-zero inputs and weights     (data-dependent power: MI355X_MICROARCH.md "DVFS give-back")
rocm-smi (sclk, socket power) is sampled in the background every ~0.25 s.   Writes gpurun_out/<tag>/inframe_gap.txt
"""
import ctypes
import os
import subprocess
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neuralrgbd_amd import ops  # noqa: E402

OUT = []


def say(*a):
    s = " ".join(str(x) for x in a)
    print(s, flush=True)
    OUT.append(s)


class Smi(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.rows, self.stop = [], False

    def run(self):
        while not self.stop:
            t = time.perf_counter()
            try:
                o = subprocess.run(["rocm-smi", "-c", "-P", "--csv"], capture_output=True, text=True, timeout=5).stdout
                self.rows.append((t, o.strip().replace("\n", " | ")))
            except Exception as e:      # noqa: BLE001
                self.rows.append((t, "rocm-smi failed: %r" % (e,)))
            time.sleep(0.25)


clk_lib = ctypes.CDLL(os.path.join(ROOT, "tools", "probes", "libclock_probe.so"))
clk_buf = torch.zeros(512, dtype=torch.int64, device="cuda")


def shader_ghz():
    """Launch the clock probe on the current stream (behind whatever was queued) and return the median clock over 256 CUs."""
    clk_lib.clock_probe(ctypes.c_void_p(clk_buf.data_ptr()), 256, 6000, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    r = clk_buf.reshape(256, 2).double().cpu()
    return float((r[:, 0] / (r[:, 1] * 10.0)).median())


def per_launch(fn, n, gap_s=0.0):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
        if gap_s:
            torch.cuda.synchronize(); time.sleep(gap_s)
    ghz = shader_ghz()
    return np.array([a.elapsed_time(b) for a, b in ev]), ghz


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r4gap"
    smi = Smi(); smi.start()
    D, H, W = 64, 192, 256
    g = torch.Generator().manual_seed(0)
    x = torch.randn(D, H, W, 64, generator=g).cuda()
    w = (torch.randn(64, 64, 3, 3, 3, generator=g) * 0.05).cuda()
    ss = torch.rand(64, 2, generator=g).cuda()
    wdw = ops.conv_wino_dw_pack(w)
    y = torch.empty_like(x)

    def layer():
        ops.conv_wino_dw(x, wdw, 64, x_ss=ss, x_relu=True)

    layer(); torch.cuda.synchronize()
    say("idle clock %.2f GHz" % shader_ghz())
    time.sleep(2.0)
    t, ghz = per_launch(layer, 5)
    say("burst      5 launches: %s ms   clock behind them %.2f GHz" % (np.round(t, 3).tolist(), ghz))
    time.sleep(2.0)
    t0 = time.perf_counter()
    t, ghz = per_launch(layer, 600)
    say("sustained  600 launches in %.2f s: first 5 %.3f | 20-40 %.3f | 100-200 %.3f | 300-600 %.3f ms (min %.3f max %.3f)   clock behind them %.2f GHz"
        % (time.perf_counter() - t0, t[:5].mean(), t[20:40].mean(), t[100:200].mean(), t[300:].mean(), t.min(), t.max(), ghz))
    say("           every 50th: " + " ".join("%.3f" % v for v in t[::50]))
    time.sleep(2.0)
    t, ghz = per_launch(layer, 100, gap_s=0.02)
    say("gapped     100 launches, 20 ms idle between: mean %.3f (min %.3f max %.3f) ms   clock %.2f GHz" % (t.mean(), t.min(), t.max(), ghz))
    xz, wz = torch.zeros_like(x), ops.conv_wino_dw_pack(torch.zeros_like(w))
    time.sleep(2.0)
    t, ghz = per_launch(lambda: ops.conv_wino_dw(xz, wz, 64, x_ss=ss, x_relu=True), 300)
    say("zeros      300 launches: first 5 %.3f | 100-300 %.3f ms   clock %.2f GHz" % (t[:5].mean(), t[100:].mean(), ghz))
    del xz, wz

    # ---- the frame (bench.py's own set-up) and the layer call bench.py's roofline_mfma times
    import neuralrgbd_amd
    from neuralrgbd_amd import camera, synth
    from neuralrgbd_amd.streaming import DepthStream
    import bench
    Hi, Wi = 768, 1024
    cam = camera.scannet_intrinsics(Wi // 4, Hi // 4)
    d_candi = np.linspace(0.1, 5.0, 64)
    model = neuralrgbd_amd.KVNET(64, cam, d_candi, 10.0, 64, None, if_refined=True, refineNet_name="DPV", t_win_r=2)
    model.load_state_dict(synth.seeded_state_dict(model, 0))
    model = model.cuda()
    ring = [tuple(t_.cuda() for t_ in synth.noise_window(i, Hi, Wi, 4)) for i in range(2)]
    kt = bench.KernelTimer(keep=lambda a, k: a[0].shape[-1] == 64 and k.get("res") is None and k.get("x_ss") is not None and not k.get("materialize"))
    ops.conv_wino_dw = kt.wrap(ops.conv_wino_dw)
    st = DepthStream(model, cam, d_candi, t_win_r=2, use_graph=True, device=torch.device("cuda"))
    for i in range(4):
        st.step(*ring[i % 2])
    torch.cuda.synchronize(); time.sleep(2.0)
    e = [torch.cuda.Event(enable_timing=True) for _ in range(61)]
    e[0].record()
    for i in range(60):
        st.step(*ring[i % 2]); e[i + 1].record()
    ghz = shader_ghz()
    ft = np.array([e[i].elapsed_time(e[i + 1]) for i in range(60)])
    say("frame      60 frames: first 3 %.2f | 10-20 %.2f | 40-60 %.2f ms   clock behind them %.2f GHz" % (ft[:3].mean(), ft[10:20].mean(), ft[40:].mean(), ghz))
    a, k = kt.last
    real = lambda: kt.fn(*a, **k)       # noqa: E731
    t, ghz = per_launch(real, 5)
    say("frame's own layer call, 5 launches right behind the frames: %s ms   clock %.2f GHz" % (np.round(t, 3).tolist(), ghz))
    time.sleep(2.0)
    t, ghz = per_launch(real, 5)
    say("frame's own layer call, 5 launches after 2 s of idle:       %s ms   clock %.2f GHz" % (np.round(t, 3).tolist(), ghz))
    t, ghz = per_launch(real, 300)
    say("frame's own layer call, 300 launches sustained: first 5 %.3f | 100-300 %.3f ms   clock %.2f GHz" % (t[:5].mean(), t[100:].mean(), ghz))
    xr = a[0]
    say("frame's layer input: %.1f %% zeros after the fused BatchNorm+ReLU prologue would be applied; |x| mean %.3f" % (
        100.0 * float(((xr * k["x_ss"][:, 0] + k["x_ss"][:, 1]) <= 0).float().mean()), float(xr.abs().mean())))
    smi.stop = True
    time.sleep(0.3)
    say("rocm-smi samples (t [s], clocks / power):")
    t00 = smi.rows[0][0] if smi.rows else 0
    for tt, row in smi.rows:
        say("  %6.2f  %s" % (tt - t00, row[-300:]))
    d = os.path.join(ROOT, "gpurun_out", tag)
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "inframe_gap.txt"), "w") as f:
        f.write("\n".join(OUT) + "\n")


if __name__ == "__main__":
    main()
