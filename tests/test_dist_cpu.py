"""The N>1 path on CPU: world_size-2 gloo process groups (no GPU needed).

Covers what bench.py --gpus N and a data-parallel training step rely on: stream sharding without a data-path
collective, the max-over-ranks timing reduction, and the bucketed gradient all-reduce (sum / world, shared
parameters reduced once, identical replicas after the step)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from neuralrgbd_amd import distributed as nd


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    r, w = nd.init_from_env("gloo")
    assert (r, w) == (rank, world)
    out = {}
    # 1. inference: independent streams, no collective; every stream owned by exactly one rank
    mine = nd.shard_streams(5, world, rank)
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    out["streams"] = gathered
    # 2. bench timing contract: MAX over ranks
    out["tmax"] = nd.max_over_ranks(1.0 + rank)
    # 3. data-parallel training step on a small module with a shared (aliased) sub-module
    torch.manual_seed(0)
    shared = torch.nn.Linear(4, 4)
    model = torch.nn.ModuleDict({"a": shared, "alias": shared, "b": torch.nn.Linear(4, 2), "unused": torch.nn.Linear(3, 3)})
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    reducer = nd.GradAllReduce(model, bucket_mb=1e-4)  # tiny buckets: several collectives
    x = torch.full((3, 4), float(rank + 1))
    loss = model["b"](model["alias"](model["a"](x))).pow(2).sum()
    loss.backward()
    local = [p.grad.clone() for p in reducer.params if p.grad is not None]
    reducer()
    out["n_params"], out["n_buckets"] = len(reducer.params), len(reducer.buckets)
    out["grad0"] = reducer.params[0].grad.numpy().copy()   # numpy: pickled by value through the queue
    out["local0"] = local[0].numpy().copy()
    opt.step()
    out["weights"] = torch.cat([p.detach().reshape(-1) for p in reducer.params]).numpy().copy()
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_world_size_2_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0]["streams"] == [[0, 2, 4], [1, 3]]
    assert res[0]["tmax"] == res[1]["tmax"] == 2.0
    assert res[0]["n_params"] == 6 and res[0]["n_buckets"] > 1          # alias counted once; unused kept
    mean = (res[0]["local0"] + res[1]["local0"]) / 2
    assert np.allclose(res[0]["grad0"], mean) and np.allclose(res[1]["grad0"], mean)
    assert np.array_equal(res[0]["weights"], res[1]["weights"])        # replicas stay identical


def test_shard_streams_partition():
    for world in (1, 2, 4, 8):
        owned = sorted(s for r in range(world) for s in nd.shard_streams(13, world, r))
        assert owned == list(range(13))
