"""Data-parallel plumbing of the hot path: one process per GPU, `torch.distributed` (backend "nccl" == RCCL over xGMI
on ROCm; "gloo" in the CPU tests).

Replaces the reference's single-process DataParallel (lib/nn/parallel/data_parallel.py:74-116): no per-step
parameter broadcast (`replicate.py:12`), no scatter/gather through GPU 0 (`_functions.py:6-86`), no Python thread per
GPU (`parallel_apply.py:50-59`).  What remains of it on this path:

  * the batch is per-image independent, so rank r of W owns images r, r+W, ...  and every
    RoI / detection belonging to them -- no data-path collective;
  * the one exchange step of a training iteration is the gradient reduction
    (`Broadcast.backward -> ReduceAddCoalesced`, `_functions.py:26-39`), here an all-reduce of flat fp32 buckets.
    The reference sums the replicas' gradients of a loss that is the MEAN over GPUs
    (`utils/training_stats.py:84`), i.e. averaged gradients == `ReduceOp.AVG` semantics.
"""
import torch
import torch.distributed as dist

BUCKET_BYTES = 64 << 20  # xGMI is point-to-point (7 links x ~153 GB/s): few, large messages per link


def world():
    """(rank, world_size) -- (0, 1) when torch.distributed is not initialised."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_range(num_items, rank=None, world_size=None):
    """Indices of the items (images) rank `rank` owns: rank, rank + W, rank + 2W, ...  Round-robin rather than
    contiguous blocks so that a dataset sorted by aspect ratio / size spreads evenly."""
    if rank is None or world_size is None:
        rank, world_size = world()
    return list(range(rank, num_items, world_size))


def shard_rois_by_image(rois, num_images, rank=None, world_size=None):
    """Split a [R,5] RoI tensor (batch_index, x1, y1, x2, y2) by image ownership.  Returns (local_rois, index) where
    local_rois carries batch indices renumbered to the rank's local image order and `index` are the rows of `rois`
    that were kept (so per-RoI outputs can be scattered back)."""
    mine = shard_range(num_images, rank, world_size)
    remap = torch.full((max(num_images, 1),), -1, dtype=torch.long, device=rois.device)
    if mine:
        remap[torch.tensor(mine, device=rois.device)] = torch.arange(len(mine), device=rois.device)
    local_batch = remap[rois[:, 0].long().clamp(0, max(num_images - 1, 0))]
    valid = (rois[:, 0] >= 0) & (rois[:, 0] < num_images)
    keep = torch.nonzero((local_batch >= 0) & valid, as_tuple=False).flatten()
    local = rois[keep].clone()
    local[:, 0] = local_batch[keep].to(rois.dtype)
    return local, keep


class GradientAllReducer(object):
    """The gradient exchange of a data-parallel step, overlapped with the backward that produces the gradients
    (replaces `Broadcast.backward -> ReduceAddCoalesced` to GPU 0, nn/parallel/_functions.py:26-39).

    Trainable parameters are packed, in the order their gradients become ready (reverse registration order: heads first,
    backbone last), into a few large flat fp32 buckets; every `p.grad` is a VIEW into its bucket, so autograd accumulates
    straight into the communication buffer -- no pack / unpack passes over the 176.5 MB payload of e2e_mask_rcnn_R-50-FPN.
    A post-accumulate hook counts a bucket's gradients down; the last one issues the bucket's asynchronous all-reduce
    (RCCL runs it on its own stream while the rest of the backward continues on the compute stream).  The reference's loss
    is a mean over GPUs (utils/training_stats.py:84): on RCCL the collective itself averages (`ReduceOp.AVG`, no extra pass
    over the payload); backends without AVG (gloo, the CPU tests) sum and `finish_step()` divides.  A parameter that got
    no gradient on ANY rank this step leaves finish_step() with `.grad = None`, as in the reference (SGD then skips its
    weight decay and momentum).

    Bucket size: xGMI is point-to-point (7 links x ~153 GB/s per GPU), a ring step moves bucket/W bytes per link, so few
    large buckets amortise the per-collective latency; 4 buckets of <= 64 MB still leave 3/4 of the payload overlappable.

        reducer = GradientAllReducer(model.parameters())
        per step:  reducer.begin_step(); loss.backward(); reducer.finish_step(); optimizer.step()
    (use `optimizer.zero_grad(set_to_none=False)` or none at all: begin_step() zero-fills the buckets and re-attaches the
    views)."""

    def __init__(self, params, bucket_bytes=BUCKET_BYTES, group=None, force=False, overlap=True, detect_unused=False):
        """`overlap=False`: the hooks only count; every bucket is reduced in finish_step() (the mode a hipGraph-captured
        backward needs: collectives stay outside the captured region -- see reduce_now()).
        `detect_unused=True`: one more (tiny) all-reduce per step tells every rank which parameters received no gradient
        on ANY rank; their `.grad` is None after finish_step(), as in the reference.  Off by default: in the shipped graphs
        every trainable parameter gets a gradient, and without the flag exchange such a parameter simply keeps an all-zero
        averaged gradient (SGD then still applies weight decay / momentum to it -- the one deviation)."""
        self.group = group
        self.overlap = overlap
        self.detect_unused = detect_unused
        self.params = [p for p in params if p.requires_grad]
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        self.active = self.world > 1 or (force and dist.is_available() and dist.is_initialized())
        self.buckets = []       # (flat, [(param, offset)])
        self.payload_bytes = sum(p.numel() for p in self.params) * 4
        self._handles = []
        self._hooks = []
        self.measure_exposed = False  # bench.py: time the compute stream's stall in finish_step() with events
        self._exposed_events = []
        if not self.active:
            return
        backend = dist.get_backend(group)
        self._avg_in_collective = backend == "nccl" and hasattr(dist.ReduceOp, "AVG")
        self._op = dist.ReduceOp.AVG if self._avg_in_collective else dist.ReduceOp.SUM
        order = list(reversed(self.params))
        cur, size = [], 0
        groups = []
        for p in order:
            nbytes = p.numel() * 4
            if cur and size + nbytes > bucket_bytes:
                groups.append(cur)
                cur, size = [], 0
            cur.append(p)
            size += nbytes
        if cur:
            groups.append(cur)
        for members in groups:
            total = sum(p.numel() for p in members)
            flat = torch.zeros(total, dtype=torch.float32, device=members[0].device)
            slots, off = [], 0
            for p in members:
                if p.dtype != torch.float32:
                    raise TypeError("GradientAllReducer expects fp32 parameters (master weights)")
                slots.append((p, off))
                off += p.numel()
            self.buckets.append((flat, slots))
        self._pending = [0] * len(self.buckets)
        self._next = 0          # first bucket not yet handed to the backend this step
        # one flag per parameter: "a gradient arrived on this rank this step"; reduced (max) with the buckets so that every
        # rank knows which parameters got no gradient anywhere
        self._index = {}
        for b, (_, slots) in enumerate(self.buckets):
            for p, _ in slots:
                self._index[id(p)] = len(self._index)
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(b, self._index[id(p)])))
        self._fired_host = [False] * len(self._index)

    def _make_hook(self, b, k):
        def hook(_param):
            self._fired_host[k] = True
            self._pending[b] -= 1
            if self._pending[b] == 0 and self.overlap:
                self._launch_ready()
        return hook

    def _launch_ready(self):
        """Buckets are reduced strictly in bucket order on every rank: a bucket that is complete waits for its
        predecessors.  (A rank on which a parameter got no gradient never completes that bucket from its hooks; launching
        whatever is ready would then pair different buckets across ranks -- found by the two-rank test.)"""
        while self._next < len(self.buckets) and self._pending[self._next] == 0:
            self._launch(self._next)
            self._next += 1

    def _launch(self, b):
        flat = self.buckets[b][0]
        self._handles.append(dist.all_reduce(flat, op=self._op, group=self.group, async_op=True))

    def begin_step(self):
        if not self.active:
            return
        self._handles = []
        self._next = 0
        self._fired_host = [False] * len(self._index)
        for b, (flat, slots) in enumerate(self.buckets):
            flat.zero_()
            self._pending[b] = len(slots)
            for p, off in slots:
                p.grad = flat[off:off + p.numel()].view_as(p)

    def reduce_now(self):
        """All buckets, now, on the current stream (the graph-replay step: the backward that filled the buckets was a
        hipGraph launch, no hook ran).  Returns the number of collectives."""
        if not self.active:
            return 0
        assert not self.detect_unused, "detect_unused reads the autograd hooks' flags: incompatible with a replayed backward"
        handles = [dist.all_reduce(flat, op=self._op, group=self.group, async_op=True) for flat, _ in self.buckets]
        for h in handles:
            h.wait()
        return len(handles)

    def average_(self):
        """Sums -> means, in place (captured into the optimizer graph in replay mode); nothing to do when the collective
        averaged."""
        if self.active and self.world > 1 and not self._avg_in_collective:
            for flat, _ in self.buckets:
                flat.div_(self.world)

    def finish_step(self):
        """Wait for every bucket (a bucket whose hooks never all fired -- a parameter without gradient this step -- is
        reduced now, so that all ranks issue the same collectives) and average.  Returns the number of collectives."""
        if not self.active:
            return 0
        for b in range(self._next, len(self.buckets)):  # what the hooks did not launch, in bucket order
            self._pending[b] = 0
            self._launch(b)
        self._next = len(self.buckets)
        timed = self.measure_exposed and self.buckets[0][0].is_cuda
        if timed:  # between the two events the compute stream does nothing but wait for the collectives
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        for h in self._handles:
            h.wait()
        if timed:
            ev1.record()
            self._exposed_events.append((ev0, ev1))
        if self.world > 1 and not self._avg_in_collective:
            for flat, _ in self.buckets:
                flat.div_(self.world)
        if self.detect_unused:
            # every rank issues this collective: agree on the parameters that got no gradient on any rank
            if not any(self._fired_host):
                # no hook ran at all: the backward was replayed from a hipGraph (hooks do not run on replay) or begin_step()
                # was skipped -- every flag would read "unused" and the whole model's gradients would be dropped silently
                raise RuntimeError("GradientAllReducer(detect_unused=True): no gradient hook fired in this step; "
                                   "detect_unused needs an eagerly executed backward between begin_step() and finish_step()")
            flags = torch.tensor([1.0 if f else 0.0 for f in self._fired_host], device=self.buckets[0][0].device)
            if self.world > 1:
                dist.all_reduce(flags, op=dist.ReduceOp.MAX, group=self.group)
            dead = set(k for k, v in enumerate(flags.tolist()) if v == 0.0)
            for p in self.params:
                if self._index[id(p)] in dead:
                    p.grad = None
        return len(self._handles)

    def exposed_ms(self):
        """Mean time per step the compute stream waited for the gradient exchange (communication the backward did not
        hide), over the steps recorded while `measure_exposed` was set; None without records.  Synchronises."""
        if not self._exposed_events:
            return None
        torch.cuda.synchronize()
        ms = [a.elapsed_time(b) for a, b in self._exposed_events]
        self._exposed_events = []
        return sum(ms) / len(ms)

    def close(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []


def max_over_ranks(seconds, device=None):
    """The step time the job sees: the slowest rank's (bench.py takes MAX over ranks)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
