"""K-Net stack at config B: residual layers fused (today) vs split (materialise pass + plain conv).  HIP events, steady state."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from neuralrgbd_amd import nets
torch.manual_seed(0)
net = nets.KalmanGainNet(16, feature_dim=64).cuda()
D, H, W = (64, 192, 256) if len(sys.argv) < 2 else tuple(int(v) for v in sys.argv[1].split("x"))
vol = torch.randn(D, H, W, 16, device="cuda")
def t(n=12, warm=6):
    with torch.no_grad():
        for _ in range(warm): out = net.forward_channels_last(vol)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): out = net.forward_channels_last(vol)
        e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, out
a, ya = t()
nets.KalmanGainNet._split_residual = True
b, yb = t()
print("K-Net %dx%dx%d: fused residual layers %.2f ms, split %.2f ms; max|d| %.2e (|gain|max %.2f)" % (D, H, W, a, b, (ya - yb).abs().max().item(), ya.abs().max().item()))
