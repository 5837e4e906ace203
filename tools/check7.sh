cd $GRAFT_REPO_ROOT
O=gpurun_out/c7; mkdir -p $O
timeout 200 python -m pytest tests/test_gpu_train.py -m gpu -q -x -s -k "knet_training_path" 2>&1 | grep -E "parity|assert|Error|passed|failed" | head > $O/t.txt
# the probe uses torch.manual_seed(SEED) for the input; the test a Generator with seed 3 — scan a few seeds of the test's form
python - > $O/seeds.txt 2>&1 <<'PY'
import copy, torch, sys
sys.path.insert(0, '.')
from neuralrgbd_amd import nets, synth
dev = "cuda:0"
net = nets.KalmanGainNet(16, feature_dim=64)
net.load_state_dict(synth.seeded_state_dict(net, 5))
cpu = copy.deepcopy(net).double()
D, H, W = 4, 12, 24
for seed in range(1, 13):
    vol = torch.randn(1, 16, D, H, W, generator=torch.Generator().manual_seed(seed))
    c = copy.deepcopy(cpu)
    c(vol.double())[0, 0].square().sum().backward()
    gold = {n: p.grad.float() for n, p in c.named_parameters()}
    a = copy.deepcopy(net).to(dev); b = copy.deepcopy(net).to(dev)
    a(vol.to(dev))[0, 0].square().sum().backward()
    b.forward_channels_last_autograd(vol[0].permute(1, 2, 3, 0).contiguous().to(dev)).square().sum().backward()
    w = lambda m: max(((p.grad.cpu() - gold[n]).abs().max() / gold[n].abs().max()).item() for n, p in m.named_parameters())
    print("seed %2d: vendor modules %.2e   hand-written (Winograd) %.2e" % (seed, w(a), w(b)))
PY
cat $O/t.txt $O/seeds.txt
