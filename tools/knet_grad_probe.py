"""Which side of the flaky K-Net gradient comparison is wrong?  Both GPU paths vs a float64 CPU autograd reference."""
import copy, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from neuralrgbd_amd import nets, synth
dev = "cuda:0"
net = nets.KalmanGainNet(16, feature_dim=64)
net.load_state_dict(synth.seeded_state_dict(net, 5))
cpu = copy.deepcopy(net).double()
D, H, W = 4, 12, 24
torch.manual_seed(int(os.environ.get("SEED", "1")))
vol = torch.randn(1, 16, D, H, W)
cpu(vol.double())[0, 0].square().sum().backward()
gold = {n: p.grad.float() for n, p in cpu.named_parameters()}
for trial in range(1):
    a = copy.deepcopy(net).to(dev); b = copy.deepcopy(net).to(dev)
    a(vol.to(dev))[0, 0].square().sum().backward()                                   # torch modules / MIOpen
    b.forward_channels_last_autograd(vol[0].permute(1, 2, 3, 0).contiguous().to(dev)).square().sum().backward()
    wa = max(((p.grad.cpu() - gold[n]).abs().max() / gold[n].abs().max()).item() for n, p in a.named_parameters())
    wb = max(((p.grad.cpu() - gold[n]).abs().max() / gold[n].abs().max()).item() for n, p in b.named_parameters())
    na = max(a.named_parameters(), key=lambda t: ((t[1].grad.cpu() - gold[t[0]]).abs().max() / gold[t[0]].abs().max()).item())[0]
    print("trial %d: worst rel. error vs fp64 CPU  modules/MIOpen %.2e (%s)   hand-written kernels %.2e" % (trial, wa, na, wb))
