"""RCCL on the GPU box.  The builder has one GPU per run, so this is a world-size-1 process group over the `nccl` (= RCCL)
backend: it proves that RCCL initialises on the MI355X node, that the bucketed gradient all-reduce of
neuralrgbd_amd.distributed runs its collectives on device buffers from backward hooks, and that bench.py's N > 1 timing helper
(barrier + MAX over ranks) works on CUDA tensors.  The multi-rank logic itself is covered on CPU (gloo, world size 2,
tests/test_dist_cpu.py); the 1/2/4/8-GPU curve is the driver's."""
import socket

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_rccl_single_rank_gradient_allreduce_and_max_over_ranks():
    from neuralrgbd_amd import distributed as nd
    assert not dist.is_initialized()
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % _free_port(), rank=0, world_size=1,
                            device_id=torch.device(DEV))
    try:
        t = torch.arange(8, dtype=torch.float32, device=DEV)
        dist.all_reduce(t)
        dist.barrier()
        assert torch.equal(t.cpu(), torch.arange(8, dtype=torch.float32))
        assert nd.max_over_ranks(1.25, device=torch.device(DEV)) == 1.25

        torch.manual_seed(0)
        model = torch.nn.Sequential(torch.nn.Linear(64, 128), torch.nn.ReLU(), torch.nn.Linear(128, 32)).to(DEV)
        ref = [p.detach().clone() for p in model.parameters()]
        reducer = nd.GradAllReduce(model, bucket_mb=0.01)        # several small buckets -> several collectives
        assert len(reducer.buckets) >= 2
        x = torch.randn(16, 64, device=DEV)
        reducer.prepare()
        model(x).square().mean().backward()
        reducer()
        torch.cuda.synchronize()
        # world size 1: the reduced gradient is the local gradient
        twin = torch.nn.Sequential(torch.nn.Linear(64, 128), torch.nn.ReLU(), torch.nn.Linear(128, 32)).to(DEV)
        for p, r in zip(twin.parameters(), ref):
            p.data.copy_(r)
        twin(x).square().mean().backward()
        for p, q in zip(model.parameters(), twin.parameters()):
            assert torch.allclose(p.grad, q.grad, rtol=1e-6, atol=1e-7)
    finally:
        dist.destroy_process_group()
