"""CPU baseline at n = 1 and n = 32 threads (SURVEY.md §8d asks for both): one update-branch frame of the oracle per setting
(after a first-frame call that creates the filter state), configs S and B.  python tools/cpu_n1_probe.py [S B]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import neuralrgbd_amd
from neuralrgbd_amd import camera, synth
from oracle import cpu_oracle, kvnet_oracle

for cid in sys.argv[1:] or ["S", "B"]:
    cfg = bench.CONFIGS[cid]
    H, W, D = cfg["H"], cfg["W"], cfg["D"]
    cam = camera.scannet_intrinsics(W // 4, H // 4)
    d_candi = np.linspace(cfg["d_min"], cfg["d_max"], D)
    model = neuralrgbd_amd.KVNET(64, cam, d_candi, 10.0, 64, None, if_refined=True, refineNet_name="DPV", t_win_r=2)
    sd = synth.seeded_state_dict(model, 0)
    w1, w2 = synth.noise_window(1, H, W), synth.noise_window(2, H, W)
    torch.set_num_threads(32); cpu_oracle.set_threads(32)
    o1 = kvnet_oracle.step(sd, *w1, cam, d_candi, 10.0, None)
    for n in (32, 1):
        torch.set_num_threads(n); cpu_oracle.set_threads(n)
        t0 = time.time()
        kvnet_oracle.step(sd, *w2, cam, d_candi, 10.0, o1[3])
        dt = time.time() - t0
        print("config %s  threads %2d  update frame %.1f s = %.4f frames/s" % (cid, n, dt, 1.0 / dt), flush=True)
