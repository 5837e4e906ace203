import os, sys, time, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import neuralrgbd_amd
from neuralrgbd_amd import camera, synth
from oracle import kvnet_oracle as ko, cpu_oracle as co
H,W,D=256,384,64
cam=camera.scannet_intrinsics(W//4,H//4); d=np.linspace(.1,5,D)
m=neuralrgbd_amd.KVNET(64,cam,d,10.,64,None); sd=synth.seeded_state_dict(m,0)
w1,w2=synth.noise_window(1,H,W),synth.noise_window(2,H,W)
torch.set_num_threads(32); co.set_threads(32)
o1=ko.step(sd,*w1,cam,d,10.,None)
for n in (16,32,64,128):
    torch.set_num_threads(n); co.set_threads(n)
    t0=time.time(); ko.step(sd,*w2,cam,d,10.,o1[3]); print(n,'threads: S update frame', round(time.time()-t0,2),'s', flush=True)
