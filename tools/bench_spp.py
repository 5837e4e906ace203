"""spp_concat at the config-B grid (5 frames, 192 x 256, 64 + 128 + 4 x 32 channels), HIP events."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuralrgbd_amd import ops
N, h, w = 5, 192, 256
quarter = torch.randn(N, h, w, 64, device="cuda"); deep = torch.randn(N, h, w, 128, device="cuda")
br = [(torch.randn(N, h // k, w // k, 32, device="cuda"), torch.randn(32, 2, device="cuda")) for k in (8, 16, 32, 64)]
for _ in range(5): out = ops.spp_concat(quarter, deep, br)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): out = ops.spp_concat(quarter, deep, br)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 50
print("spp_concat %dx%dx%d: %.1f us  (%.2f TB/s of %d MB)" % (N, h, w, ms * 1e3, (out.numel() + quarter.numel() + deep.numel()) * 4 / ms / 1e9, (out.numel() + quarter.numel() + deep.numel()) * 4 >> 20))
