"""Multi-process CPU tests (gloo, world_size 2) of the data-parallel plumbing (detectron_pytorch_amd/parallel.py):
image sharding without a data-path collective, and the bucketed gradient all-reduce that replaces the reference's
Broadcast.backward -> ReduceAddCoalesced (lib/nn/parallel/_functions.py:26-39)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from detectron_pytorch_amd import parallel


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world_size, port, fn, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    try:
        ret[rank] = fn(rank, world_size)
    finally:
        dist.destroy_process_group()


def _run(fn, world_size=2):
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world_size, port, fn, ret)) for r in range(world_size)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    return [ret[r] for r in range(world_size)]


def _shard_job(rank, world_size):
    rois = torch.tensor([[0, 1, 1, 5, 5], [1, 2, 2, 6, 6], [2, 3, 3, 7, 7], [3, 0, 0, 4, 4], [1, 9, 9, 12, 12],
                         [7, 0, 0, 1, 1]], dtype=torch.float32)  # last row: no such image
    local, keep = parallel.shard_rois_by_image(rois, num_images=4)
    t = parallel.max_over_ranks(0.5 + rank)
    return parallel.shard_range(5), local.tolist(), keep.tolist(), t


def test_images_and_rois_shard_without_overlap():
    out = _run(_shard_job)
    assert out[0][0] == [0, 2, 4] and out[1][0] == [1, 3]
    # rank 0 owns images 0, 2 -> local batch 0, 1 ; rank 1 owns images 1, 3
    assert out[0][2] == [0, 2] and out[1][2] == [1, 3, 4]
    assert [r[0] for r in out[0][1]] == [0.0, 1.0] and [r[0] for r in out[1][1]] == [0.0, 1.0, 0.0]
    assert sorted(out[0][2] + out[1][2]) == [0, 1, 2, 3, 4]  # every valid RoI exactly once, the invalid one nowhere
    assert out[0][3] == out[1][3] == 1.5  # max over ranks


def _grad_job(rank, world_size):
    torch.manual_seed(0)
    params = [torch.nn.Parameter(torch.zeros(n)) for n in (5, 1000, 3, 70000)]
    frozen = torch.nn.Parameter(torch.zeros(4), requires_grad=False)
    for i, p in enumerate(params):
        p.grad = torch.full_like(p, float((rank + 1) * (i + 1)))
    params[2].grad = None if rank == 1 else params[2].grad  # a parameter unused on one rank
    n_coll = parallel.allreduce_gradients(params + [frozen], bucket_bytes=8192)
    return n_coll, [float(p.grad[0]) for p in params], [bool(torch.all(p.grad == p.grad[0])) for p in params]


def test_bucketed_gradient_allreduce_averages_like_the_reference():
    out = _run(_grad_job)
    for n_coll, firsts, uniform in out:
        # buckets: [5 + 1000 floats] | [3] is packed with what fits ... the 70000-float tensor alone exceeds a bucket
        assert 2 <= n_coll <= 4
        assert all(uniform)
        np.testing.assert_allclose(firsts, [1.5 * 1, 1.5 * 2, 0.5 * 3, 1.5 * 4])  # mean over ranks; None counts as 0
    assert out[0] == out[1]


def test_single_process_is_a_no_op():
    p = torch.nn.Parameter(torch.ones(3))
    p.grad = torch.ones(3)
    assert parallel.allreduce_gradients([p]) == 0 and parallel.world() == (0, 1)
    assert parallel.shard_range(3, 0, 1) == [0, 1, 2]
