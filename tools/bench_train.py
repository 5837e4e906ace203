#!/usr/bin/env python
"""Training-step throughput (BASELINE config 4 shape: ScanNet 384x256 image, grid 96x64x64, N = 1 per GPU).

    python tools/bench_train.py [--iters 6]
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/bench_train.py

One iteration = neuralrgbd_amd.train_step.train(): forward under autograd, 4 NLL terms, backward (fused cost-volume
backward kernel + torch for the convolutions), bucketed gradient all-reduce when WORLD_SIZE > 1, Adam step, PREDICT.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=6)
    ap.add_argument("--graph", action="store_true", help="replay the captured hipGraph of the iteration (1 GPU)")
    args = ap.parse_args()
    import neuralrgbd_amd
    from neuralrgbd_amd import camera, distributed as nd, synth
    from neuralrgbd_amd.train_step import train
    rank, world = nd.init_from_env()
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    H, W, D = 256, 384, 64
    cam = camera.scannet_intrinsics(W // 4, H // 4)
    d_candi = np.linspace(0.1, 5, D)
    model = neuralrgbd_amd.KVNET(64, cam, d_candi, 10.0, 64, None, if_refined=True, refineNet_name="DPV", t_win_r=2)
    model.load_state_dict(synth.seeded_state_dict(model, 0))
    model = model.to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-5, betas=(.9, .999), capturable=args.graph)   # local_train_scanNet.sh
    reducer = nd.GradAllReduce(model) if world > 1 else None
    rng = np.random.RandomState(rank)
    pred, times = None, []
    tg = None
    if args.graph:
        from neuralrgbd_amd.train_step import TrainGraph
        assert world == 1, "--graph is the single-GPU form"
        tg = TrainGraph(model, opt, 2, d_candi, cam)
    for it in range(args.iters + (4 if args.graph else 2)):
        r, s, p = synth.noise_window(100 * rank + it, H, W)
        ref = [{"img": r, "dmap": torch.from_numpy(rng.randint(0, D, (1, H // 4, W // 4))),
                "dmap_imgsize_digit": torch.from_numpy(rng.randint(0, D, (1, H, W)))}]
        src = [[{"img": s[0, v:v + 1]} for v in range(4)]]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if tg is not None and it >= 3:      # eager iterations first: filter state, vendor find-mode, optimizer state
            loss, pred = tg.step(r.to(dev), s.to(dev), p.to(dev), ref[0]["dmap"].to(dev), ref[0]["dmap_imgsize_digit"].to(dev), pred)
        else:
            _, pred, loss, _, _ = train(world, model, opt, 2, d_candi, ref, src, p, pred, [cam], grad_reducer=reducer)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    steady = times[5:] if args.graph else times[2:]
    dt = nd.max_over_ranks(float(np.mean(steady)), device=dev)
    if rank == 0:
        print("train step (grid 96x64x64, N=1/GPU, %d GPU): %.1f ms/iteration, %.2f windows/s aggregate, loss %.3f, "
              "gradient message %.2f MB" % (world, 1e3 * dt, world / dt, float(loss),
                                            4e-6 * (reducer.numel if reducer else sum(p.numel() for p in set(model.parameters())))))


if __name__ == "__main__":
    main()
