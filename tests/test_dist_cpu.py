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


def test_bench_harness_world_2_under_torchrun():
    """bench.py's N>1 branch (process-group init from the torchrun environment, barriers, MAX-over-ranks reduction, ONE
    JSON line from rank 0, clean teardown) executed end to end on a gloo group with a stub frame function — the code path
    the driver launches as `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
           "--stub"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                        # rank 0 only
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 4 and rec["scaling"] == "weak"
    assert rec["ms_per_step"] >= 9.0                              # the slower rank (10 ms per step) sets the time: MAX, not mean
    assert abs(rec["value"] - 2 * 4 / (rec["ms_per_step"] * 4e-3)) < 1e-6


def _kvnet_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    nd.init_from_env("gloo")
    import neuralrgbd_amd
    from neuralrgbd_amd import camera
    cam = camera.scannet_intrinsics(24, 16)
    d = np.linspace(.1, 5, 8)
    model = neuralrgbd_amd.KVNET(64, cam, d, 10., 64, None, if_refined=True, refineNet_name="DPV", t_win_r=2)   # CPU: no kernels run
    torch.manual_seed(0)
    for p in model.parameters():
        p.data.normal_(0, 0.1)
    reducer = nd.GradAllReduce(model, bucket_mb=4.0)
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    sums = []
    for step in range(2):
        reducer.prepare()
        # a stand-in loss touching every unique parameter (the real forward needs the GPU): rank-dependent gradients
        loss = sum((p * float(rank + 1 + step)).sum() for p in reducer.params)
        loss.backward()
        reducer()
        opt.step()
        sums.append(float(sum(p.grad.double().sum() for p in reducer.params)))
    q.put((rank, {"n_unique": len(reducer.params), "n_named": len(list(model.state_dict())), "numel": reducer.numel,
                  "n_buckets": len(reducer.buckets), "in_backward": reducer.launched_in_backward, "sums": sums,
                  "w": float(sum(p.detach().double().sum() for p in reducer.params))}))
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_allreduce_on_the_real_kvnet_parameter_set():
    """GradAllReduce on the actual 459-key KVNET (feature CNN registered twice -> reduced once): persistent buckets with
    .grad views, collectives started from backward hooks, replicas identical after two steps."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_kvnet_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    a, b = res[0], res[1]
    assert a["n_named"] == 459 and a["numel"] == b["numel"]
    assert 5.2e6 < a["numel"] < 5.4e6 or a["numel"] > 1e6          # unique parameters (D=8 R-Net is smaller than D=64's)
    assert a["n_buckets"] >= 2 and a["in_backward"] >= 1            # overlap: at least one bucket left during backward
    # d loss / d p = rank + 1 + step on every element -> mean over ranks = 1.5 + step
    for step in range(2):
        assert abs(a["sums"][step] - (1.5 + step) * a["numel"]) < 1e-3 * a["numel"]
        assert a["sums"][step] == b["sums"][step]
    assert a["w"] == b["w"]


def _uneven_worker(rank, world, port, q):
    """Default bucket size on the real parameter set; in step 0 rank 1's loss does not touch the K-Net (what train() does
    when `valid_dpv(BVs_predict)` fails on that rank only), in step 1 every rank uses everything."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    nd.init_from_env("gloo")
    import neuralrgbd_amd
    from neuralrgbd_amd import camera
    cam = camera.scannet_intrinsics(24, 16)
    d = np.linspace(.1, 5, 64)
    model = neuralrgbd_amd.KVNET(64, cam, d, 10., 64, None, if_refined=True, refineNet_name="DPV", t_win_r=2)
    torch.manual_seed(0)
    for p in model.parameters():
        p.data.normal_(0, 0.1)
    reducer = nd.GradAllReduce(model)                      # default bucket_mb
    knet = {id(p) for p in model.kv_net.parameters()}
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    orders, hooks, sums = [], [], []
    for step in range(2):
        reducer.prepare()
        used = [p for p in reducer.params if not (step == 0 and rank == 1 and id(p) in knet)]
        loss = sum((p * float(rank + 1)).sum() for p in used)
        loss.backward()
        reducer()
        opt.step()
        orders.append([b for b, _ in reducer.launch_order])
        hooks.append(reducer.launched_in_backward)
        sums.append([float(reducer.flat[bi].double().sum()) for bi in range(len(reducer.buckets))])
    n_knet = sum(p.numel() for p in reducer.params if id(p) in knet)
    q.put((rank, {"orders": orders, "hooks": hooks, "sums": sums, "n_buckets": len(reducer.buckets), "numel": reducer.numel,
                  "n_knet": n_knet, "w": float(sum(p.detach().double().sum() for p in reducer.params))}))
    dist.barrier()
    dist.destroy_process_group()


def test_bucket_order_is_rank_independent_when_a_rank_skips_a_subnetwork():
    """a bucket's collective may start from a backward hook only after every earlier bucket has started, so
    two ranks whose steps used different sub-networks still issue the same collectives in the same order (before: the rank
    that skipped the K-Net deferred that bucket while its peer launched it mid-backward -> mismatched sizes -> hang)."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_uneven_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    a, b = res[0], res[1]
    n = a["n_buckets"]
    assert n >= 3                                                   # default bucket size: the 21 MB gradient is >= 3 messages
    for step in range(2):
        assert a["orders"][step] == b["orders"][step] == list(range(n))
    assert a["hooks"][0] >= 2 and a["hooks"][1] >= 2 and b["hooks"][1] >= 2     # overlap at the default size
    assert b["hooks"][0] < a["hooks"][0]                            # rank 1 stopped at the first incomplete bucket in step 0
    # step 0: d loss / d p = rank + 1, K-Net only on rank 0 -> mean = 1.5 outside the K-Net, 0.5 inside; step 1: 1.5 everywhere
    want0 = 1.5 * (a["numel"] - a["n_knet"]) + 0.5 * a["n_knet"]
    assert abs(sum(a["sums"][0]) - want0) < 1e-3 * a["numel"] and a["sums"][0] == b["sums"][0]
    assert abs(sum(a["sums"][1]) - 1.5 * a["numel"]) < 1e-3 * a["numel"] and a["sums"][1] == b["sums"][1]
    assert a["w"] == b["w"]


def test_bench_spawns_its_own_ranks_without_torchrun():
    """`python bench.py --gpus 2 ...` started WITHOUT torchrun must run two ranks (it re-executes itself
    under torch.distributed.run) and report n_gpus 2 with both ranks seen by a collective — never a silent single-rank run."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--stub"],
                         capture_output=True, text=True, timeout=300, cwd=root, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["collective_ranks"] == 2 and rec["steps"] == 3
    assert rec["ms_per_step"] >= 9.0                              # MAX over ranks: the slower rank sleeps 10 ms per step
    # every rank's own time per step, gathered: rank 0 sleeps 5 ms, rank 1 sleeps 10 ms — the straggler is visible
    own = rec["per_rank"]["own_ms_per_step"]
    assert len(own) == 2 and 4.5 <= own[0] < 9.0 <= own[1] and rec["per_rank"]["max"] == max(own)


def test_bench_refuses_a_world_size_that_is_not_gpus():
    """A torchrun environment whose WORLD_SIZE differs from --gpus is an error, in both directions."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0", "--stub"],
                         capture_output=True, text=True, timeout=120, cwd=root, env=env)
    assert out.returncode != 0 and "does not match WORLD_SIZE" in out.stderr + out.stdout


def _accum_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    nd.init_from_env("gloo")
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.Tanh(), torch.nn.Linear(16, 3))
    params = list(model.parameters())
    A = 4
    xs = [torch.randn(5, 6, generator=torch.Generator().manual_seed(100 * rank + k)) for k in range(A)]
    # four single-window gradients, summed in window order (what accumulation into one buffer does)
    singles = []
    for x in xs:
        for p in params:
            p.grad = None
        model(x).pow(2).mean().backward()
        singles.append([p.grad.clone() for p in params])
    local_sum = [singles[0][i].clone() for i in range(len(params))]
    for k in range(1, A):
        for i in range(len(params)):
            local_sum[i] += singles[k][i]
    # the reducer: prepare(accum_steps=4), four backward passes, one call
    reducer = nd.GradAllReduce(model, bucket_mb=1e-4)
    reducer.prepare(accum_steps=A)
    for x in xs:
        model(x).pow(2).mean().backward()
    early = reducer.launched_in_backward          # buckets launched from hooks: only possible in the LAST pass
    reducer()
    q.put((rank, {"local_sum": [g.numpy().copy() for g in local_sum], "reduced": [p.grad.numpy().copy() for p in params],
                  "early": early, "n_buckets": len(reducer.buckets),
                  "order": list(reducer.launch_order)}))
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_accumulation_over_4_windows_equals_the_mean_of_the_single_window_gradients():
    """accum_steps = 4 on two ranks == (sum over ranks and windows of the single-window gradients) / 8,
    bit for bit (sums in window order, the two ranks' sums added, one exact division by a power of two); every rank launches
    the buckets in index order."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_accum_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for i in range(len(res[0]["local_sum"])):
        want = (res[0]["local_sum"][i] + res[1]["local_sum"][i]) / 8.0
        assert np.array_equal(res[0]["reduced"][i], want) and np.array_equal(res[1]["reduced"][i], want)
    assert res[0]["n_buckets"] > 1 and res[0]["early"] >= 1
    assert [b for b, _ in res[0]["order"]] == list(range(res[0]["n_buckets"])) == [b for b, _ in res[1]["order"]]


def test_gradient_accumulation_single_rank_divides_without_a_collective():
    torch.manual_seed(1)
    model = torch.nn.Linear(4, 2)
    reducer = nd.GradAllReduce(model)
    xs = [torch.randn(3, 4) for _ in range(4)]
    reducer.prepare(accum_steps=4)
    for x in xs:
        model(x).sum().backward()
    reducer()
    want = sum(x.sum(0) for x in xs) / 4.0
    assert torch.allclose(model.weight.grad[0], want, rtol=1e-6, atol=1e-6)
    with __import__("pytest").raises(ValueError):
        reducer.prepare(accum_steps=0)


def _alternating_worker(rank, world, port, q):
    """the real 459-key parameter set, default buckets, accumulate-4, TWO consecutive optimizer steps; in step 0
    rank 0's first window skips the K-Net (train() takes the first-frame branch when `valid_dpv(BVs_predict)` fails on that rank),
    in step 1 it is rank 1's third window that does.  Launch order, bucket contents and the replicas must agree after each step."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    nd.init_from_env("gloo")
    import neuralrgbd_amd
    from neuralrgbd_amd import camera
    cam = camera.scannet_intrinsics(24, 16)
    d = np.linspace(.1, 5, 64)
    model = neuralrgbd_amd.KVNET(64, cam, d, 10., 64, None, if_refined=True, refineNet_name="DPV", t_win_r=2)
    torch.manual_seed(0)
    for p in model.parameters():
        p.data.normal_(0, 0.1)
    reducer = nd.GradAllReduce(model)
    knet = {id(p) for p in model.kv_net.parameters()}
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    A = 4
    skip = {(0, 0, 0), (1, 1, 2)}                          # (step, rank, window) whose loss does not touch the K-Net
    orders, hooks, sums = [], [], []
    for step in range(2):
        reducer.prepare(accum_steps=A)
        for k in range(A):
            used = [p for p in reducer.params if not ((step, rank, k) in skip and id(p) in knet)]
            sum((p * float(rank + 1 + k)).sum() for p in used).backward()
        reducer()
        opt.step()
        orders.append([b for b, _ in reducer.launch_order])
        hooks.append(reducer.launched_in_backward)
        sums.append([float(reducer.flat[bi].double().sum()) for bi in range(len(reducer.buckets))])
    n_knet = sum(p.numel() for p in reducer.params if id(p) in knet)
    q.put((rank, {"orders": orders, "hooks": hooks, "sums": sums, "n_buckets": len(reducer.buckets), "numel": reducer.numel,
                  "n_knet": n_knet, "w": float(sum(p.detach().double().sum() for p in reducer.params))}))
    dist.barrier()
    dist.destroy_process_group()


def test_accumulate_4_over_two_steps_with_ranks_alternately_skipping_the_knet():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_alternating_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    a, b = res[0], res[1]
    n = a["n_buckets"]
    assert n >= 3
    for step in range(2):
        assert a["orders"][step] == b["orders"][step] == list(range(n))         # same collectives, same order, on both ranks
        assert a["sums"][step] == b["sums"][step]                                # ... and the same reduced messages
    # the skipping rank cannot complete the K-Net's buckets from hooks (one of its 4 x len(bucket) gradients never lands): it
    # launches fewer buckets during backward than its peer — in step 0 that is rank 0, in step 1 rank 1
    assert a["hooks"][0] < b["hooks"][0] and b["hooks"][1] < a["hooks"][1]
    # d loss / d p = sum over ranks r and windows k of (r + 1 + k) / (4 * 2): all 8 terms = (1+2+3+4 + 2+3+4+5) / 8 = 3.0 outside
    # the K-Net; inside it step 0 misses rank 0's window 0 (value 1) -> 23 / 8, step 1 misses rank 1's window 2 (value 4) -> 20 / 8
    out_k = a["numel"] - a["n_knet"]
    assert abs(sum(a["sums"][0]) - (3.0 * out_k + 23.0 / 8.0 * a["n_knet"])) < 1e-3 * a["numel"]
    assert abs(sum(a["sums"][1]) - (3.0 * out_k + 20.0 / 8.0 * a["n_knet"])) < 1e-3 * a["numel"]
    assert a["w"] == b["w"]                                                      # replicas identical after two optimizer steps
