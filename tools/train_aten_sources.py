#!/usr/bin/env python
"""Which host code launches the ATen (non-nrgbd) kernels of one eager training iteration: torch.profiler with shapes and Python
stacks, grouped by (op, input shapes, innermost neuralrgbd_amd frame).  python tools/train_aten_sources.py [op substring ...]"""
import os, sys, collections
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import neuralrgbd_amd
from neuralrgbd_amd import camera, synth
from neuralrgbd_amd.train_step import train

want = sys.argv[1:] or ["copy_", "fill_", "add", "zero_", "cat", "flip", "mean", "sum", "index"]
dev = torch.device("cuda", 0)
H, W, D = 256, 384, 64
cam = camera.scannet_intrinsics(W // 4, H // 4)
d_candi = np.linspace(0.1, 5, D)
model = neuralrgbd_amd.KVNET(64, cam, d_candi, 10.0, 64, None, if_refined=True, refineNet_name="DPV", t_win_r=2)
model.load_state_dict(synth.seeded_state_dict(model, 0))
model = model.to(dev)
opt = torch.optim.Adam(model.parameters(), lr=1e-5, betas=(.9, .999))
rng = np.random.RandomState(0)
pred = None


def one(it):
    global pred
    r, s, p = synth.noise_window(it, H, W)
    ref = [{"img": r, "dmap": torch.from_numpy(rng.randint(0, D, (1, H // 4, W // 4))),
            "dmap_imgsize_digit": torch.from_numpy(rng.randint(0, D, (1, H, W)))}]
    src = [[{"img": s[0, v:v + 1]} for v in range(4)]]
    _, pred, loss, _, _ = train(1, model, opt, 2, d_candi, ref, src, p, pred, [cam])


for it in range(3):
    one(it)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    one(3)
    torch.cuda.synchronize()
groups = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    name = ev.name
    if not name.startswith("aten::") or not any(w in name for w in want):
        continue
    dt = getattr(ev, "self_device_time_total", 0) or getattr(ev, "self_cuda_time_total", 0)
    if dt <= 0:
        continue
    frame = "?"
    for fr in ev.stack or []:
        if "neuralrgbd_amd/" in fr and "ops.py" not in fr:
            frame = fr.split("neuralrgbd_amd/")[-1]
            break
    else:
        for fr in ev.stack or []:
            if "torch/autograd" in fr or "torch/optim" in fr:
                frame = fr.split("site-packages/")[-1][:70]
                break
    key = (name, str(ev.input_shapes)[:90], frame[:80])
    groups[key][0] += 1
    groups[key][1] += dt
tot = sum(v[1] for v in groups.values())
print("ATen ops matching %s with device time: %.0f us in one iteration" % (want, tot))
for k, v in sorted(groups.items(), key=lambda kv: -kv[1][1])[:60]:
    print("%8.1f us %4d x  %-22s %-90s %s" % (v[1], v[0], k[0], k[1], k[2]))
