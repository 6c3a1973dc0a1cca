"""Multi-process CPU tests (gloo, world_size 2) of the data-parallel plumbing (detectron_pytorch_amd/parallel.py):
image sharding without a data-path collective, and `GradientAllReducer` -- the class every training step uses -- which
replaces the reference's Broadcast.backward -> ReduceAddCoalesced (lib/nn/parallel/_functions.py:26-39): gradient views
into flat buckets, all-reduce launched from autograd hooks, a bucket whose hooks did not all fire, the hook-less mode of
a captured backward, replicas that stay bit-identical."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from detectron_pytorch_amd import parallel

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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
        p.join(180)
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


class _Net(torch.nn.Module):
    """Two heads on a trunk; `head_b` is only used when asked (a head without work on one rank), `spare` never."""

    def __init__(self):
        super().__init__()
        self.trunk = torch.nn.Linear(32, 300)
        self.head_a = torch.nn.Linear(300, 7)
        self.head_b = torch.nn.Linear(300, 70)   # 21 070 floats: a bucket of its own at 32 KB
        self.spare = torch.nn.Linear(5, 5)
        self.frozen = torch.nn.Parameter(torch.ones(4), requires_grad=False)

    def forward(self, x, use_b):
        h = torch.relu(self.trunk(x))
        out = self.head_a(h).square().mean()
        if use_b:
            out = out + self.head_b(h).abs().mean()
        return out


def _reducer_job(rank, world_size, overlap, detect_unused):
    torch.manual_seed(5)
    net = _Net()
    ref = _Net()
    ref.load_state_dict(net.state_dict())
    red = parallel.GradientAllReducer(net.parameters(), bucket_bytes=32 << 10, overlap=overlap, detect_unused=detect_unused)
    assert red.active and red.world == 2 and len(red.buckets) >= 2
    opt = torch.optim.SGD(net.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-3)
    opt_ref = torch.optim.SGD(ref.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-3)
    xs = [torch.randn(8, 32, generator=torch.Generator().manual_seed(10 * r + 1)) for r in range(world_size)]
    log = []
    for step in range(3):
        use_b = [step != 1 or r == 0 for r in range(world_size)]  # step 1: head_b has no work on rank 1
        opt.zero_grad(set_to_none=True)
        red.begin_step()
        net(xs[rank], use_b[rank]).backward()
        # every gradient lives in a bucket: the view, not a copy
        for flat, slots in red.buckets:
            for p, off in slots:
                assert p.grad.data_ptr() == flat[off:off + p.numel()].data_ptr()
        n_coll = red.finish_step()
        assert n_coll == len(red.buckets)
        # the reference semantics on one process: mean over ranks of the per-rank losses (training_stats.py:84)
        opt_ref.zero_grad(set_to_none=True)
        (sum(ref(xs[r], use_b[r]) for r in range(world_size)) / world_size).backward()
        for (name, p), q in zip(net.named_parameters(), ref.parameters()):
            if not p.requires_grad:
                continue
            if q.grad is None:  # `spare`: no gradient on any rank
                assert (p.grad is None) if detect_unused else bool((p.grad == 0).all()), name
            else:
                torch.testing.assert_close(p.grad, q.grad, rtol=1e-6, atol=1e-7, msg=name)
        opt.step()
        opt_ref.step()
        log.append(torch.cat([p.detach().reshape(-1) for p in net.parameters()]).double().sum().item())
    red.close()
    return log


def _hooks_job(rank, world_size):
    return _reducer_job(rank, world_size, overlap=True, detect_unused=False)


def _no_overlap_job(rank, world_size):
    return _reducer_job(rank, world_size, overlap=False, detect_unused=True)


def test_reducer_hook_mode_matches_the_reference_mean_and_replicas_stay_identical():
    out = _run(_hooks_job)
    assert out[0] == out[1]  # bit-identical parameter checksums after every one of the three SGD steps


def test_reducer_without_overlap_and_with_unused_parameter_detection():
    out = _run(_no_overlap_job)
    assert out[0] == out[1]


def _graph_mode_job(rank, world_size):
    """reduce_now() / average_(): the exchange of a step whose backward ran without hooks (a replayed hipGraph)."""
    torch.manual_seed(1)
    net = torch.nn.Linear(16, 4)
    red = parallel.GradientAllReducer(net.parameters(), bucket_bytes=1 << 20, overlap=False)
    red.begin_step()
    for p in net.parameters():
        p.grad.fill_(float(rank + 1))   # what a captured backward would have left in the bucket views
    n = red.reduce_now()
    red.average_()
    return n, [float(p.grad.reshape(-1)[0]) for p in net.parameters()]


def test_reduce_now_averages_bucket_views():
    out = _run(_graph_mode_job)
    assert out[0] == out[1] == (1, [1.5, 1.5])


def test_single_process_is_a_no_op():
    net = torch.nn.Linear(3, 3)
    red = parallel.GradientAllReducer(net.parameters())
    assert not red.active and red.finish_step() == 0 and red.reduce_now() == 0 and parallel.world() == (0, 1)
    assert parallel.shard_range(3, 0, 1) == [0, 1, 2]


def test_bench_selftest_two_ranks_on_gloo():
    """bench.py --gpus 2 spawns its own ranks under torch.distributed.run; on gloo the whole rank / reducer / timing /
    JSON plumbing runs without a GPU and checks that the replicas did not diverge."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--selftest-cpu", "--gpus", "2", "--steps", "3",
                          "--warmup", "1"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["metric"] == "selftest" and line["n_gpus"] == 2 and line["collectives_per_step"] >= 1


def test_cpu_quota_helper_keeps_the_intra_op_pool_inside_the_container_budget(monkeypatch):
    """hostcpu: the cgroup quota is parsed (v2 'quota period' / 'max'), and respect_cpu_quota only ever lowers the pool."""
    import builtins
    import io

    import torch

    from detectron_pytorch_amd import hostcpu

    real_open = builtins.open

    def fake_open(content):
        def opener(path, *a, **k):
            if path == "/sys/fs/cgroup/cpu.max":
                return io.StringIO(content)
            return real_open(path, *a, **k)
        return opener

    monkeypatch.setattr(builtins, "open", fake_open("1600000 100000\n"))
    assert hostcpu.cpu_quota() == 16.0
    monkeypatch.setattr(builtins, "open", fake_open("max 100000\n"))
    assert hostcpu.cpu_quota() is None
    monkeypatch.setattr(builtins, "open", fake_open("800000 100000\n"))
    monkeypatch.delenv("OMP_NUM_THREADS", raising=False)
    before = torch.get_num_threads()
    try:
        n = hostcpu.respect_cpu_quota()
        assert 1 <= n <= max(2, before) and n <= max(before, 2)
        assert n == min(before, 2)
    finally:
        torch.set_num_threads(before)
