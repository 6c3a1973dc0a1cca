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
import pytest
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
    assert red.active and red.world == world_size and len(red.buckets) >= 2
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


@pytest.mark.parametrize("world_size", [4, 8])
def test_reducer_at_four_and_eight_ranks(world_size):
    """The rank counts BASELINE's configs 4 / 5 run at.  Hook mode with in-order bucket launches (a head without work on
    every rank but one in step 1), then the no-overlap mode with unused-parameter detection: gradients equal the
    single-process mean over the ranks' losses, the replicas' parameter checksums stay bit-identical over three SGD steps."""
    for job in (_hooks_job, _no_overlap_job):
        out = _run(job, world_size)
        assert all(o == out[0] for o in out)


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


def _sync_and_losses_job(rank, world_size):
    """sync_parameters / assert_replicas_equal / reduce_losses: what is left of the reference's per-forward parameter
    broadcast (nn/parallel/replicate.py:12) and of its mean-over-GPUs logging (utils/training_stats.py:84)."""
    torch.manual_seed(11)   # every rank builds the same net ...
    net = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.BatchNorm1d(16), torch.nn.Linear(16, 4))
    net[1].running_mean.add_(0.25)                      # ... a buffer too
    reference = [t.clone() for t in parallel._state_tensors(net)]
    if rank == 1:                                       # ... and rank 1 then diverges: weights, a buffer, an integer buffer
        with torch.no_grad():
            net[0].weight.add_(torch.randn(16, 8))
            net[2].bias.mul_(3.0)
            net[1].running_var.add_(1.0)
            net[1].num_batches_tracked.add_(7)
    diverged = False
    try:
        parallel.assert_replicas_equal(net)
    except RuntimeError:
        diverged = True                                 # raised on BOTH ranks (the verdict travels in the collective)
    sent = parallel.sync_parameters(net, src=0, bucket_bytes=256)   # tiny buckets: several broadcasts per dtype
    parallel.assert_replicas_equal(net)
    same_as_rank0 = all(torch.equal(a, b) for a, b in zip(parallel._state_tensors(net), reference))
    # the losses of this rank's shard; the logged value is the mean over the ranks
    ret = {"losses": {"loss_cls": torch.tensor(1.0 + rank), "loss_bbox": torch.tensor([0.5 * (rank + 1)])},
           "metrics": {"accuracy_cls": 0.25 + 0.5 * rank}, "total_loss": torch.tensor(3.0 * (rank + 1))}
    logged = parallel.reduce_losses(ret)
    return diverged, same_as_rank0, sent, logged, int(parallel.replica_checksum(net).item())


@pytest.mark.parametrize("world_size", [2, 4, 8])
def test_sync_parameters_makes_a_perturbed_replica_bit_identical_and_losses_are_averaged(world_size):
    out = _run(_sync_and_losses_job, world_size)
    assert all(o[0] for o in out)                        # the divergence (of rank 1 alone) was seen on every rank
    assert all(o[1] for o in out)                        # afterwards all hold rank 0's state, bit for bit
    assert all(o[2] == out[0][2] > 0 and o[4] == out[0][4] for o in out)
    mean_rank = (world_size - 1) / 2.0
    want = {"losses": {"loss_cls": 1.0 + mean_rank, "loss_bbox": 0.5 * (mean_rank + 1)},
            "metrics": {"accuracy_cls": 0.25 + 0.5 * mean_rank}, "total_loss": 3.0 * (mean_rank + 1)}
    assert all(o[3] == want for o in out)                # = the mean of the shard values (training_stats.py:84)


def _no_hook_job(rank, world_size):
    """detect_unused with a step in which no hook ran on ONE rank only: every rank must raise (after the collective) instead
    of one rank raising in front of it and the other hanging in the all-reduce."""
    torch.manual_seed(2)
    net = torch.nn.Linear(4, 2)
    red = parallel.GradientAllReducer(net.parameters(), overlap=False, detect_unused=True)
    red.begin_step()
    if rank == 0:
        net(torch.ones(3, 4)).sum().backward()
    try:
        red.finish_step()
    except RuntimeError as exc:
        return "no gradient hook fired" in str(exc)
    return False


@pytest.mark.parametrize("world_size", [2, 4, 8])
def test_detect_unused_without_hooks_on_one_rank_raises_on_every_rank(world_size):
    assert _run(_no_hook_job, world_size) == [True] * world_size


def test_replica_helpers_without_a_process_group():
    net = torch.nn.Linear(3, 3)
    assert parallel.sync_parameters(net) == 0 and parallel.assert_replicas_equal(net)
    assert parallel.reduce_losses({"a": torch.tensor(2.0), "b": {"c": 1}}) == {"a": 2.0, "b": {"c": 1.0}}
    c0 = int(parallel.replica_checksum(net).item())
    with torch.no_grad():
        net.weight[0, 0] += 1e-3
    assert int(parallel.replica_checksum(net).item()) != c0
    # differences INSIDE one tensor must not cancel: two elements swapped, and +d / -d on two integer elements
    buf = torch.nn.Module()
    buf.register_buffer("a", torch.tensor([5, 9, 2, 7], dtype=torch.int64))
    c1 = int(parallel.replica_checksum(buf).item())
    buf.a[0], buf.a[1] = 9, 5
    assert int(parallel.replica_checksum(buf).item()) != c1
    buf.a[0], buf.a[1] = 5 + 3, 9 - 3
    assert int(parallel.replica_checksum(buf).item()) != c1


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


def test_bench_selftest_under_an_external_launcher():
    """The form the measurement driver uses for N > 1: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...` -- bench.py must take the ranks it is given (no second
    spawn) and rank 0 must print exactly one JSON line."""
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                          "--selftest-cpu", "--gpus", "2", "--steps", "2", "--warmup", "1"],
                         cwd=ROOT, env=dict(os.environ), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines
    line = json.loads(lines[0])
    assert line["metric"] == "selftest" and line["n_gpus"] == 2 and line["steps"] == 2


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


# ---- the REAL training step at world size 2 (rcnn.train.train_step + GradientAllReducer; CPU backends of tests/cpu_backend.py) ----
def _rcnn_setup():
    """A small Mask R-CNN R-50-FPN job: seeded model, two images of 128 x 160 with two gt boxes each, the data layer's RPN
    target blobs per image, fixed sampling priorities (so that no random stream has to line up between processes)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import cpu_backend  # noqa: F401  (import check: the spawned workers need the same path)
    from detectron_pytorch_amd.rcnn import config, data as rdata, model as rmodel

    torch.set_num_threads(1)      # identical summation orders in every process
    from detectron_pytorch_amd.rcnn import targets

    if not getattr(targets.label_proposals, "_sliced", False):
        # the CPU backend collects however many proposals survive NMS: cut the (long) fixed priority vector to fit
        inner = targets.label_proposals

        def label_proposals(cfg, rois, gt_boxes, gt_classes, gt_image, scales, priority, *a, **k):
            return inner(cfg, rois, gt_boxes, gt_classes, gt_image, scales, priority[:gt_boxes.size(0) + rois.size(0)], *a, **k)

        label_proposals._sliced = True
        targets.label_proposals = label_proposals
    cfg = config.mask_rcnn_r50_fpn()
    cfg.merge(dict(TRAIN=dict(BATCH_SIZE_PER_IM=32, RPN_PRE_NMS_TOP_N=100, RPN_POST_NMS_TOP_N=60)))
    torch.manual_seed(cfg.RNG_SEED)
    net = rmodel.GeneralizedRCNN(cfg).train()
    h, w, g = 128, 160, 2
    rng = np.random.RandomState(23)
    shards = []
    for i in range(2):
        bw, bh = rng.uniform(20, 110, g), rng.uniform(20, 90, g)
        x1, y1 = rng.uniform(0, w - 1 - bw), rng.uniform(0, h - 1 - bh)
        boxes = np.stack([x1, y1, x1 + bw, y1 + bh], 1).astype(np.float32)
        classes = rng.randint(1, 81, g).astype(np.int32)
        entry = dict(height=h, width=w, boxes=boxes, gt_classes=classes, is_crowd=np.zeros(g, bool))
        blobs = rdata.add_rpn_blobs(cfg, [entry], [1.0], np.random.RandomState(31 + i))
        shards.append(dict(
            data=torch.from_numpy((rng.randn(1, 3, h, w) * 50).astype(np.float32)),
            im_info=torch.from_numpy(blobs["im_info"]),
            roidb={"gt_boxes": torch.from_numpy(boxes), "gt_classes": torch.from_numpy(classes).long(),
                   "gt_image": torch.zeros(g, dtype=torch.long)},
            rpn_t={k: torch.from_numpy(v) for k, v in blobs.items() if k.startswith("rpn_")},
            priority=torch.from_numpy(rng.permutation(g + 600).astype(np.float32))))
    return cfg, net, shards


def _rcnn_rank_job(rank, world_size):
    import cpu_backend
    from detectron_pytorch_amd.rcnn import train as rtrain

    cfg, net, shards = _rcnn_setup()
    s = shards[rank]                                   # a different image on every rank
    opt = rtrain.make_optimizer(net, cfg, lr=2e-3)
    reducer = parallel.GradientAllReducer(net.parameters(), bucket_bytes=16 << 20)
    assert reducer.active and len(reducer.buckets) >= 3
    losses = []
    with cpu_backend.cpu_ops(net):
        for _ in range(3):
            ret = rtrain.train_step(net, opt, s["data"], s["im_info"], s["roidb"], s["rpn_t"], reducer=reducer,
                                    priority=s["priority"])
            losses.append(float(ret["total_loss"]))
    digest = torch.cat([p.detach().reshape(-1)[:64] for p in net.parameters() if p.requires_grad])
    full = {"Box_Outs.cls_score.weight": None, "Conv_Body.conv_body.res5.2.conv3.weight": None, "RPN.FPN_RPN_conv.bias": None}
    named = dict(net.named_parameters())
    return losses, digest.numpy(), {k: named[k].detach().numpy().copy() for k in full}


def test_real_training_step_replicas_stay_identical_and_equal_the_sharded_single_process_step():
    """Three iterations of rcnn.train.train_step (Mask R-CNN R-50-FPN, 44 M trainable parameters in four buckets) on two gloo
    ranks with DIFFERENT images: the replicas' parameters stay bit-identical, and equal what one process computes when it
    runs the two shards one after the other and averages their gradients -- the reference's semantics (the loss of a step is
    the mean over GPUs of per-GPU means, utils/training_stats.py:84; gradients are summed on GPU 0 by ReduceAddCoalesced,
    nn/parallel/_functions.py:26-39, and the per-GPU losses were already divided by their own normalisers)."""
    out = _run(_rcnn_rank_job)
    (l0, d0, p0), (l1, d1, p1) = out
    assert np.array_equal(d0, d1), "replicas diverged"
    for k in p0:
        assert np.array_equal(p0[k], p1[k]), k
    assert l0 != l1, "the ranks were meant to see different images"
    # one process, both shards, gradients averaged by hand
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import cpu_backend
    from detectron_pytorch_amd.rcnn import train as rtrain
    from detectron_pytorch_amd.rcnn.model import total_loss

    threads = torch.get_num_threads()
    try:
        cfg, net, shards = _rcnn_setup()
        opt = rtrain.make_optimizer(net, cfg, lr=2e-3)
        params = [p for p in net.parameters() if p.requires_grad]
        seen = []
        with cpu_backend.cpu_ops(net):
            for _ in range(3):
                opt.zero_grad(set_to_none=True)
                grads, step_losses = None, []
                for s in shards:
                    for p in params:
                        p.grad = None
                    ret = net(s["data"], s["im_info"], roidb=s["roidb"], rpn_targets=s["rpn_t"], priority=s["priority"])
                    loss = total_loss(ret)
                    loss.backward()
                    step_losses.append(float(loss.detach()))
                    g = [p.grad if p.grad is not None else torch.zeros_like(p) for p in params]
                    grads = g if grads is None else [a + b for a, b in zip(grads, g)]
                for p, g in zip(params, grads):
                    p.grad = g / 2
                opt.step()
                seen.append(step_losses)
        digest = torch.cat([p.detach().reshape(-1)[:64] for p in params]).numpy()
        named = dict(net.named_parameters())
    finally:
        torch.set_num_threads(threads)
    assert [s[0] for s in seen] == l0 and [s[1] for s in seen] == l1, "per-rank losses differ from the shard losses"
    assert np.array_equal(digest, d0), "the replicas do not equal the sharded single-process step"
    for k in p0:
        assert np.array_equal(named[k].detach().numpy(), p0[k]), k
