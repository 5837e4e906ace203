"""The whole K-Net stack (nets.KalmanGainNet.forward_channels_last) at a K-Net grid with the product library or an experimental
A/B build of it (python -c "from neuralrgbd_amd import build; build.build_variant('noident', ['-DNRGBD_DW_IDENT=0'])"; NRGBD_EXP_LIB=
its path): run both in one gpurun call to compare on the same chip.  HIP events, steady state."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuralrgbd_amd import _lib
if os.environ.get("NRGBD_EXP_LIB"):
    _lib.LIB_PATH = os.environ["NRGBD_EXP_LIB"]
from neuralrgbd_amd import nets
if os.environ.get("NO_SPLIT"):          # A/B: the residual layers fused (wino_dw RES variants) instead of nhwc_act + the IDENT form
    nets.KalmanGainNet._split_residual = False
if os.environ.get("NO_D4"):             # A/B: wino_dw.hip (F(2,3) along depth) for the 64 -> 64 layers instead of wino_dw4.hip
    nets.KalmanGainNet._depth_f43 = False
if os.environ.get("D4_L0"):             # A/B: the 16 -> 64 first layer on wino_dw4.hip too
    nets.KalmanGainNet._depth_f43_cin = (16, 64)
if os.environ.get("NO_CLAMP"):            # A/B of the clamped-FMA ReLU form: the plain form everywhere
    nets._relu_unit = lambda owner, bn, count: 0.0
torch.manual_seed(0)
net = nets.KalmanGainNet(16, feature_dim=64).cuda()
D, H, W = (64, 192, 256) if len(sys.argv) < 2 else tuple(int(v) for v in sys.argv[1].split("x"))
vol = torch.randn(D, H, W, 16, device="cuda")
with torch.no_grad():
    for _ in range(8):
        out = net.forward_channels_last(vol)
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            out = net.forward_channels_last(vol)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10)
print("K-Net %dx%dx%d with %s%s: %s ms per pass; checksum %.6f" % (D, H, W, os.path.basename(_lib.LIB_PATH), (" (no clamp form)" if os.environ.get("NO_CLAMP") else "") + (" (fused residual layers)" if os.environ.get("NO_SPLIT") else ""), " ".join("%.2f" % t for t in ts), out.double().abs().mean().item()))
