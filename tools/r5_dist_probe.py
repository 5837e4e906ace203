"""Is the eager training step bit-reproducible run to run, and does the gradient reducer (without / with RCCL world-size-1
collectives) change its result?  One-off probe behind tests/test_gpu_dist.py's first-contact test."""
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import neuralrgbd_amd
from neuralrgbd_amd import camera, distributed as nd, synth
from neuralrgbd_amd.optim import FusedAdam
from neuralrgbd_amd.train_step import train
DEV = "cuda:0"
H, W, D = 256, 256, 8
cam = camera.scannet_intrinsics(W // 4, H // 4)
d_candi = np.linspace(0.1, 5, D)
rng = np.random.RandomState(5)
labels = [(torch.from_numpy(rng.randint(0, D, (1, H // 4, W // 4))).to(DEV), torch.from_numpy(rng.randint(0, D, (1, H, W))).to(DEV)) for _ in range(8)]


def make():
    m = neuralrgbd_amd.KVNET(64, cam, d_candi, 10.0, 64, None, if_refined=True, refineNet_name="DPV", t_win_r=2)
    m.load_state_dict(synth.seeded_state_dict(m, 0))
    return m.to(DEV)


def run(reducer_kind, opt_kind="fused", steps=3):
    m = make()
    opt = FusedAdam(m.parameters(), lr=1e-4) if opt_kind == "fused" else torch.optim.SGD(m.parameters(), lr=1e-3)
    red = None
    if reducer_kind == "plain":
        red = nd.GradAllReduce(m, bucket_mb=2.0)
    elif reducer_kind == "rccl":
        red = nd.GradAllReduce(m, bucket_mb=2.0, always_collective=True)
    elif reducer_kind == "rccl_nooverlap":
        red = nd.GradAllReduce(m, bucket_mb=2.0, always_collective=True, overlap=False)
    pred = None
    for it in range(steps):
        r, s, p = synth.noise_window(3000 + it, H, W)
        dm, dmf = labels[it]
        _, pred, loss, _, _ = train(1, m, opt, 2, d_candi, [{"img": r.to(DEV), "dmap": dm, "dmap_imgsize_digit": dmf}],
                                    [[{"img": s[0, v:v + 1].to(DEV)} for v in range(4)]], p.to(DEV), pred, [cam], grad_reducer=red)
    torch.cuda.synchronize()
    return m, float(loss)


def diff(a, b):
    n, worst, name = 0, 0.0, ""
    for (k, p), q in zip(a.named_parameters(), b.parameters()):
        d = (p - q).abs().max().item()
        if d > 0:
            n += 1
        if d > worst:
            worst, name = d, k
    return n, worst, name


for optk in ("fused", "sgd"):
    a, la = run(None, optk)
    b, lb = run(None, optk)
    print(optk, "twin vs twin:", diff(a, b), la, lb)
    c, lc = run("plain", optk)
    print(optk, "reducer (no group) vs twin:", diff(c, a), lc)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29533", rank=0, world_size=1, device_id=torch.device(DEV))
for optk in ("fused", "sgd"):
    a, la = run(None, optk)
    d_, ld = run("rccl", optk)
    print(optk, "reducer (RCCL, hooks) vs twin:", diff(d_, a), ld)
    e, le = run("rccl_nooverlap", optk)
    print(optk, "reducer (RCCL, no overlap) vs twin:", diff(e, a), le)
    f, lf = run("rccl", optk, steps=1)
    g, lg = run(None, optk, steps=1)
    print(optk, "1 step: RCCL hooks vs twin:", diff(f, g))
dist.destroy_process_group()
