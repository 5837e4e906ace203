"""Timing of the K-Net 16->64 first layer (conv3d_mfma<16>) at the config-B grid."""
import torch, sys
sys.path.insert(0,'.')
from neuralrgbd_amd import ops
D,H,W=64,192,256
x=torch.randn(D,H,W,16,device='cuda'); w=torch.randn(64,16,3,3,3,device='cuda')*0.05
wp=ops.conv3d_pack_weights(w); y=torch.empty(D,H,W,64,device='cuda')
f=lambda: ops.conv3d(x,wp,out=y)
f(); torch.cuda.synchronize()
e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): f()
e1.record(); torch.cuda.synchronize()
print("conv3d 16->64 %.3f ms"%(e0.elapsed_time(e1)/10))
