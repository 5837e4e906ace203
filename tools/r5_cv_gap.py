"""Why does the fused sampling kernel take 0.235 ms in bench.py's back-to-back re-launch loop and 0.206 ms where it sits in the frame?
The frame's own call (same tensors) timed per launch with HIP events: back to back, with an unrelated kernel between launches, with the
frame's neighbours (pack before, warp volume behind), and inside consecutive eager frames."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import neuralrgbd_amd
from neuralrgbd_amd import camera, ops, synth

dev = torch.device("cuda:0")
H, W, D, V = 768, 1024, 64, 4
cam = camera.scannet_intrinsics(W // 4, H // 4)
d_candi = np.linspace(0.1, 5.0, D)
model = neuralrgbd_amd.KVNET(64, cam, d_candi, 10.0, 64, None, if_refined=True, refineNet_name="DPV", t_win_r=2)
model.load_state_dict(synth.seeded_state_dict(model, 0))
model = model.to(dev)
ring = [tuple(t.to(dev) for t in synth.noise_window(i, H, W, V)) for i in range(2)]
last = {}
orig = ops.costvol
times = []


def wrapped(*a, **k):
    last["a"], last["k"] = a, k
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); out = orig(*a, **k); e1.record()
    times.append((e0, e1))
    return out


ops.costvol = wrapped
from neuralrgbd_amd.streaming import DepthStream
st = DepthStream(model, cam, d_candi, t_win_r=2, use_graph=False, device=dev)
for i in range(3):
    st.step(*ring[i % 2])
torch.cuda.synchronize(); times.clear()
for i in range(8):
    st.step(*ring[i % 2])
torch.cuda.synchronize()
print("in consecutive eager frames: %s us" % " ".join("%.0f" % (1e3 * a.elapsed_time(b)) for a, b in times))
a, k = last["a"], last["k"]


def per_launch(n, between=None):
    ev = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); orig(*a, **k); e1.record(); ev.append((e0, e1))
        if between is not None:
            between()
    torch.cuda.synchronize()
    v = [1e3 * x.elapsed_time(y) for x, y in ev]
    return "mean %.0f  (first 5: %s; last 5: %s)" % (sum(v[5:]) / len(v[5:]), " ".join("%.0f" % t for t in v[:5]), " ".join("%.0f" % t for t in v[-5:]))


print("back to back, 40 launches:            ", per_launch(40))
big = torch.empty(64 * 1024 * 1024, device=dev)
print("a 256 MB fill between launches:       ", per_launch(40, lambda: big.zero_()))
small = torch.empty(1024, device=dev)
print("a tiny kernel between launches:       ", per_launch(40, lambda: small.zero_()))
x = torch.randn(8192, 8192, device=dev)
print("a 1.1 TFLOP fp32 GEMM between launches:", per_launch(20, lambda: torch.mm(x, x)))
tex_feats = torch.randn(5, 64, H // 4, W // 4, device=dev)
frames = torch.randn(5, 3, H, W, device=dev)
print("pack_nhwc between launches:           ", per_launch(40, lambda: ops.pack_nhwc(tex_feats, frames)))
