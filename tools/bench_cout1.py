"""Timing of the K-Net 64->1 layer (conv3d_cout1: tap-projection GEMM + LDS tap sum) at the config-B grid."""
import torch, sys
sys.path.insert(0,'.')
from neuralrgbd_amd import ops
D,H,W=64,192,256
x=torch.randn(D,H,W,64,device='cuda'); ss=torch.randn(64,2,device='cuda'); w=torch.randn(27,64,device='cuda')*0.05
f=lambda: ops.conv3d_cout1(x,w,x_ss=ss,x_relu=True)
f(); torch.cuda.synchronize()
e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): f()
e1.record(); torch.cuda.synchronize()
print("cout1 %.3f ms"%(e0.elapsed_time(e1)/10))
