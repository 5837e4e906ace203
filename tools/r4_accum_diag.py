import os, sys, copy
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import test_gpu_train as T
from neuralrgbd_amd.train_step import train
from neuralrgbd_amd.test_step import test as infer
model, cam, d_candi, window, (H, W, D) = T._accum_setup(5)
wins = [window(i) for i in range(4)]
with torch.no_grad():
    ref, src, p = wins[0]
    bv = infer(model, d_candi, [cam], 2, [{"img": ref["img"]}], [src], p, None)[1]
class NoStep:
    def __init__(s, params): s.params = list(params)
    def zero_grad(s):
        for q in s.params: q.grad = None
    def step(s): pass
names = [n for n, _ in model.named_parameters()]
def grad_of(win, accum_twice=False):
    ref, src, p = win
    o = NoStep(model.parameters())
    if accum_twice:
        train(1, model, o, 2, d_candi, [ref, ref], [src, src], torch.cat([p, p], 0), [bv, bv], [cam], accum_steps=2)
    else:
        train(1, model, o, 2, d_candi, [ref], [src], p, bv, [cam])
    return [q.grad.clone() for q in model.parameters()]
g1 = grad_of(wins[1]); g2 = grad_of(wins[1])
def report(tag, a, b):
    rows = sorted(((float((x - y).abs().max()), float(y.abs().max()), n) for x, y, n in zip(a, b, names)), reverse=True)[:6]
    print(tag, ["%s %.2e/%.2e" % (n, d, s) for d, s, n in rows])
report("run-to-run single:", g1, g2)
g3 = grad_of(wins[1], accum_twice=True)       # same window twice, mean = the single gradient
report("accum(2x same) vs single:", g3, g1)
# two DIFFERENT windows, each with its own state
with torch.no_grad():
    bvs = [infer(model, d_candi, [cam], 2, [{"img": w_[0]["img"]}], [w_[1]], w_[2], None)[1] for w_ in wins[:2]]
def single(win, b):
    ref, src, p = win
    train(1, model, NoStep(model.parameters()), 2, d_candi, [ref], [src], p, b, [cam])
    return [q.grad.clone() for q in model.parameters()]
ga, gb = single(wins[2], bvs[0]), single(wins[3], bvs[1])
train(1, model, NoStep(model.parameters()), 2, d_candi, [wins[2][0], wins[3][0]], [wins[2][1], wins[3][1]],
      torch.cat([wins[2][2], wins[3][2]], 0), bvs, [cam], accum_steps=2)
gacc = [q.grad.clone() for q in model.parameters()]
report("accum(2 different) vs mean of singles:", gacc, [(a + b) / 2 for a, b in zip(ga, gb)])
report("accum(2 different) vs first single:", gacc, ga)
report("accum(2 different) vs second single:", gacc, gb)
