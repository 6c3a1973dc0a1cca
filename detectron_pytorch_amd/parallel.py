"""Data-parallel plumbing of the hot path: one process per GPU, `torch.distributed` (backend "nccl" == RCCL over xGMI
on ROCm; "gloo" in the CPU tests).

Replaces the reference's single-process DataParallel (lib/nn/parallel/data_parallel.py:74-116): no per-step
parameter broadcast (`replicate.py:12`), no scatter/gather through GPU 0 (`_functions.py:6-86`), no Python thread per
GPU (`parallel_apply.py:50-59`).  What remains of it on this path:

  * the minibatch's way onto the device (`Scatter`, `_functions.py:62-83`): `MinibatchFeeder`, a copy stream of the rank's own;
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
        if self.detect_unused:
            raise RuntimeError("GradientAllReducer.reduce_now(): detect_unused reads the autograd hooks' flags, and a "
                               "replayed (hipGraph) backward runs no hook")
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
            # Every rank issues this collective: agree on the parameters that got no gradient on any rank.  The last element
            # carries "no hook ran at all on some rank" (MAX): the backward was replayed from a hipGraph (hooks do not run on
            # replay) or begin_step() was skipped -- every flag of that rank would read "unused" and gradients would be
            # dropped silently.  It travels IN the collective, so that every rank raises after it; raising before it on the
            # one rank that noticed would leave the others blocked in the all-reduce.
            none_fired = 0.0 if any(self._fired_host) else 1.0
            flags = torch.tensor([1.0 if f else 0.0 for f in self._fired_host] + [none_fired],
                                 device=self.buckets[0][0].device)
            if self.world > 1:
                dist.all_reduce(flags, op=dist.ReduceOp.MAX, group=self.group)
            flags = flags.tolist()
            if flags[-1] != 0.0:
                raise RuntimeError("GradientAllReducer(detect_unused=True): no gradient hook fired in this step on at least "
                                   "one rank; detect_unused needs an eagerly executed backward between begin_step() and "
                                   "finish_step()")
            dead = set(k for k, v in enumerate(flags[:-1]) if v == 0.0)
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


class MinibatchFeeder(object):
    """A rank's minibatch from pinned host memory into the RESIDENT device blobs the step reads, through a copy stream.

    Replaces the input half of the reference's scatter (`_functions.py:62-83`: `comm.scatter(input, gpus, ..., streams)`
    on a background stream per GPU, the main stream waits for it; called from `data_parallel.py:118-130` every forward):
    there the host runs ahead of the GPU, so the copy of step k + 1 overlaps the tail of step k.  Here the blobs keep
    their addresses (a captured hipGraph reads them), so the minibatch travels as ONE flat pinned buffer -> ONE flat
    staging buffer on the copy stream while step k computes, and `commit()` moves the pieces into the live blobs
    device-to-device on the step's stream: 73 MB of H2D at PCIe rate leave the step's timeline, 146 MB of HBM traffic
    enter it.

        feeder = MinibatchFeeder(live_tensors)
        fill(feeder.host_blobs()); feeder.prefetch()          # step 0's data
        for k in range(steps):
            feeder.commit()                                   # step k's data is live (stream-ordered; the host does not wait)
            feeder.wait_host_free(); fill(feeder.host_blobs()); feeder.prefetch()   # step k + 1's data, under step k
            step()
    """

    ALIGN = 256

    def __init__(self, live):
        self.live = list(live)
        assert self.live and all(t.is_cuda for t in self.live), "MinibatchFeeder: device tensors expected"
        dev = self.live[0].device
        offsets, total = [], 0
        for t in self.live:
            assert t.is_contiguous() or t.is_contiguous(memory_format=torch.channels_last), "MinibatchFeeder: dense blobs expected"
            offsets.append(total)
            total += (t.numel() * t.element_size() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self._flat_dev = torch.empty(total, dtype=torch.uint8, device=dev)
        self._flat_host = torch.empty(total, dtype=torch.uint8).pin_memory()

        def views(flat):
            return [flat[o:o + t.numel() * t.element_size()].view(t.dtype).as_strided(t.shape, t.stride())
                    for o, t in zip(offsets, self.live)]

        self.staging, self._host = views(self._flat_dev), views(self._flat_host)
        self.stream = torch.cuda.Stream(device=dev)
        self.ready = torch.cuda.Event()      # the staging buffer holds the prefetched minibatch (and the host buffer is free again)
        self.consumed = torch.cuda.Event()   # the last commit has read the staging buffer
        self._pending = False
        self.consumed.record(torch.cuda.current_stream(dev))
        self.ready.record(self.stream)

    def host_blobs(self):
        """Pinned host tensors shaped like the live blobs, views of one buffer: the loader writes the next minibatch here
        (after `wait_host_free()` if a prefetch may still be reading them)."""
        return self._host

    def wait_host_free(self):
        self.ready.synchronize()

    def prefetch(self, host=None):
        """Start the copy of the next minibatch: `host` (tensors shaped like the live blobs, copied into the pinned buffer
        first) or, without an argument, what the loader has written into `host_blobs()`."""
        assert not self._pending, "MinibatchFeeder: prefetch() twice without commit()"
        if host is not None:
            host = list(host)
            assert len(host) == len(self._host), "MinibatchFeeder: %d blobs, %d expected" % (len(host), len(self._host))
            self.wait_host_free()
            for dst, src in zip(self._host, host):
                assert src.shape == dst.shape and src.dtype == dst.dtype, "MinibatchFeeder: blob shape / dtype changed"
                dst.copy_(src)
        self.stream.wait_event(self.consumed)
        with torch.cuda.stream(self.stream):
            self._flat_dev.copy_(self._flat_host, non_blocking=True)
            self.ready.record(self.stream)
        self._pending = True

    def commit(self):
        assert self._pending, "MinibatchFeeder: commit() without prefetch()"
        cur = torch.cuda.current_stream(self.live[0].device)
        cur.wait_event(self.ready)
        # one multi-tensor launch per dtype instead of a memcpy call per blob (each costs ~50 us of stream time on ROCm)
        torch._foreach_copy_(self.live, self.staging)
        self.consumed.record(cur)
        self._pending = False


def _state_tensors(model):
    """Parameters and buffers of a module (or an iterable of tensors), in registration order -- the same on every rank."""
    if isinstance(model, torch.nn.Module):
        return [p.data for p in model.parameters()] + [b.data for b in model.buffers()]
    return [t.data if isinstance(t, torch.nn.Parameter) else t for t in model]


def sync_parameters(model, src=0, group=None, bucket_bytes=BUCKET_BYTES):
    """One broadcast of every parameter and buffer from rank `src`: after it all replicas are bit-identical.

    The reference re-broadcasts the parameters from GPU 0 in EVERY forward (nn/parallel/replicate.py:12 ->
    _functions.py:6-24, 176.5 MB per step for e2e_mask_rcnn_R-50-FPN); with one process per GPU the replicas only have to
    start equal -- identical averaged gradients and a deterministic optimizer keep them equal -- so this is called once,
    after the model is built or a checkpoint is loaded (a checkpoint read on one rank only, or any rank-dependent
    initialisation, would otherwise diverge silently).  Tensors travel in flat buckets of one dtype (few large messages:
    xGMI is point-to-point).  Returns the number of bytes broadcast; a no-op returning 0 without a process group."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return 0
    sent = 0
    pending, size = [], 0

    def flush():
        nonlocal pending, size, sent
        if not pending:
            return
        flat = torch.cat([t.reshape(-1) for t in pending])
        dist.broadcast(flat, src=src, group=group)
        off = 0
        for t in pending:
            t.copy_(flat[off:off + t.numel()].view_as(t))
            off += t.numel()
        sent += flat.numel() * flat.element_size()
        pending, size = [], 0

    by_type = {}
    for t in _state_tensors(model):
        by_type.setdefault((t.dtype, t.device), []).append(t)
    for (_, _), tensors in by_type.items():
        for t in tensors:
            nbytes = t.numel() * t.element_size()
            if pending and size + nbytes > bucket_bytes:
                flush()
            pending.append(t)
            size += nbytes
        flush()
    return sent


def replica_checksum(model):
    """A 64-bit checksum of every parameter and buffer: the wrapping int64 sum of the tensors' bit patterns, every ELEMENT
    weighted by an odd multiplier of its position and of its tensor's index (differences inside a tensor -- swapped
    elements, +d here and -d there -- do not cancel, nor do swapped tensors).  Cheap enough to run every K steps: one pass
    over the state, no host copy until `.item()`."""
    total = None
    for k, t in enumerate(_state_tensors(model)):
        if t.numel() == 0:
            continue
        flat = t.contiguous().reshape(-1)
        if flat.element_size() == 4:
            bits = flat.view(torch.int32).to(torch.int64)
        elif flat.element_size() == 8:
            bits = flat.view(torch.int64)
        elif flat.element_size() == 2:
            bits = flat.view(torch.int16).to(torch.int64)
        else:
            bits = flat.view(torch.uint8).to(torch.int64)
        # odd per-element weight (Knuth's multiplicative constant over the position, wrapping int64 arithmetic)
        pos = torch.arange(bits.numel(), dtype=torch.int64, device=bits.device)
        s = (bits * ((pos * 2654435761 + (2 * k + 1)) | 1)).sum()
        total = s if total is None else total + s
    return total if total is not None else torch.zeros((), dtype=torch.int64)


def assert_replicas_equal(model, group=None):
    """Raises on every rank when the replicas' parameters / buffers differ (two all-reduces of one int64: MIN and MAX of
    replica_checksum).  The reference cannot diverge -- it re-broadcasts every forward; here this is the assertion to run
    after sync_parameters() and every K steps."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return True
    mine = replica_checksum(model).reshape(1)
    lo, hi = mine.clone(), mine.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
    if int(lo.item()) != int(hi.item()):
        raise RuntimeError("data-parallel replicas have diverged: parameter checksum %d on this rank, range [%d, %d]"
                           % (int(mine.item()), int(lo.item()), int(hi.item())))
    return True


def reduce_losses(values, group=None):
    """The ~15 loss / metric scalars of a step, averaged over the ranks with ONE all-reduce -- what the reference logs
    (utils/training_stats.py:84 takes the mean over GPUs of every loss and metric of `ret`).  `values`: a dict (of dicts)
    of 0-d / 1-element tensors or floats, e.g. the `losses` / `metrics` of a training forward.  Returns the same structure
    with plain floats.  Logging only: it is not on the step's critical path (call it after finish_step(); it synchronises
    with the host once, for the `.tolist()`)."""
    names, scalars = [], []

    def walk(prefix, v):
        if isinstance(v, dict):
            for k in v:
                walk(prefix + (k,), v[k])
        else:
            names.append(prefix)
            scalars.append(v.detach().reshape(-1)[:1].float() if isinstance(v, torch.Tensor) else torch.tensor([float(v)]))

    walk((), values)
    if not scalars:
        return {}
    device = next((t.device for t in scalars if t.is_cuda), scalars[0].device)
    stacked = torch.cat([t.to(device) for t in scalars])
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        w = dist.get_world_size(group)
        if dist.get_backend(group) == "nccl" and hasattr(dist.ReduceOp, "AVG"):
            dist.all_reduce(stacked, op=dist.ReduceOp.AVG, group=group)
        else:
            dist.all_reduce(stacked, op=dist.ReduceOp.SUM, group=group)
            stacked = stacked / w
    out = {}
    for path, v in zip(names, stacked.tolist()):
        d = out
        for k in path[:-1]:
            d = d.setdefault(k, {})
        d[path[-1]] = v
    return out


def max_over_ranks(seconds, device=None):
    """The step time the job sees: the slowest rank's (bench.py takes MAX over ranks)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
