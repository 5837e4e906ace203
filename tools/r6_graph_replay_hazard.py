"""Reproducer of the hipGraph replay hazard found in round 6 (profiles/r6_graph_replay_hazard.txt).

A twin of the model trains eagerly (torch Adam) while TrainGraph replays its captured iteration for the other copy, as
tests/test_gpu_train.py::test_split_train_graph_with_accumulation_equals_eager_accumulation does.  With ROCm's graph packet capture on
(DEBUG_CLR_GRAPH_PACKET_CAPTURE=1, the runtime's default) the replayed step 2 differs from the eager one by 0.1-0.3 % of the loss for
some candidate counts / kernel routes, deterministically; with it off (what `import neuralrgbd_amd` sets unless the caller set it) they
agree to 1e-6.     DD=<candidates: 8 | 16 | 32 | 64>  ROUTE=<1 | 0: autograd.Conv2dCL.rnet_route>  DEBUG_CLR_GRAPH_PACKET_CAPTURE=<0 | 1>
"""
import os, sys, copy, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import test_gpu_train
from test_gpu_train import _accum_setup
import inspect
src = inspect.getsource(test_gpu_train._accum_setup).replace('H, W, D = 256, 256, 8', 'H, W, D = 256, 256, int(os.environ.get("DD", "8"))')
ns = dict(test_gpu_train.__dict__); ns["os"] = os; exec(src, ns); _accum_setup = ns['_accum_setup']
from neuralrgbd_amd import distributed as nd
from neuralrgbd_amd.test_step import test as infer
from neuralrgbd_amd.train_step import TrainGraph, train
from neuralrgbd_amd import autograd as ag
ag.Conv2dCL.rnet_route = os.environ.get('ROUTE', '1') == '1'
model, cam, d_candi, window, (H, W, D) = _accum_setup(6)
A = 2
wins = [window(i) for i in range(4 * A)]
preds = []
with torch.no_grad():
    for ref, src, p in wins[:A]:
        preds.append(infer(model, d_candi, [cam], 2, [{"img": ref["img"]}], [src], p, None)[1])
twin = copy.deepcopy(model)
opt = torch.optim.Adam(model.parameters(), lr=1e-4, betas=(.9, .999), capturable=True)
opt2 = torch.optim.Adam(twin.parameters(), lr=1e-4, betas=(.9, .999), capturable=True)
red2 = nd.GradAllReduce(twin)
tg = TrainGraph(twin, opt2, 2, d_candi, cam, warmup=1, grad_reducer=red2, accum_steps=A)
pe, pg = list(preds), list(preds)
import neuralrgbd_amd.train_step as ts
if os.environ.get('NOREST'):
    _orig = ts.TrainGraph._step_windows
    def patched(self, windows):
        r = _orig(self, windows)
        self._g_rest = None
        return r
    ts.TrainGraph._step_windows = patched
names = [n for n, _ in model.named_parameters()]
for it in range(3):
    batch = wins[A * (it + 1):A * (it + 2)]
    _, pred_e, loss_e, _, _ = train(1, model, opt, 2, d_candi, [b[0] for b in batch], [b[1] for b in batch],
                                    torch.cat([b[2] for b in batch], 0), pe, [cam], accum_steps=A)
    pe = list(pred_e.split(1, 0))
    windows = [(ref["img"], torch.cat([s_["img"] for s_ in src], 0).unsqueeze(0), p, ref["dmap"], ref["dmap_imgsize_digit"], pg[k])
               for k, (ref, src, p) in enumerate(batch)]
    loss_g, pg = tg.step_windows(windows)
    torch.cuda.synchronize()
    print("step", it, float(loss_g), float(loss_e))
    worst = []
    for (n, p), (_, q) in zip(model.named_parameters(), twin.named_parameters()):
        if p.grad is None or q.grad is None: continue
        d = float((p.grad - q.grad).abs().max()); s = float(p.grad.abs().max()) + 1e-30
        worst.append((d / s, n, d, s))
    worst.sort(reverse=True)
    worst_r = [w_ for w_ in worst if w_[1].startswith('r_net')]
    for w_ in []: print("   grad rel diff %.2e  %s (abs %.2e of %.2e)" % w_)
