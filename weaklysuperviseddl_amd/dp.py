"""Data-parallel training: one process per GPU, gradient all-reduce over RCCL (xGMI).

The reference is single-process (no torch.distributed anywhere); data parallelism is the north-star's
addition: replicate the 42.0 M parameters, shard the minibatch, keep per-replica BatchNorm statistics
(the reference has no SyncBN), all-reduce(sum) the 39.6 M fp32 gradients and fold the 1/world_size into
the Adam kernel's ``grad_scale``.  Parity statement: the reduced gradient equals the mean of the
per-shard single-GPU gradients.

Design for 8 x MI355X (xGMI is point-to-point, 7 links/GPU): the flat gradient buffer of ``FlatAdam`` is
cut into a few large contiguous buckets (default 4 x ~40 MB - large enough that each collective is
bandwidth- not latency-bound per link).  Parameters are laid out in forward order, so backward fills the
buffer from its END: a bucket's all-reduce is launched (async, RCCL's own stream) from the
gradient-ready hook of the last parameter it is waiting for and overlaps the remaining backward;
``wait()`` (called from ``FlatAdam.step``) only makes the compute stream wait on those collectives.

What keeps the replicas identical (none of it is in the reference, all of it is needed once there is more
than one process):
  * ``sync_initial_state``: parameters and module buffers (BatchNorm running statistics) are broadcast from
    rank 0 when the reducer is built - a checkpoint loaded on one rank or a different seed cannot leave the
    replicas apart;
  * every rank issues the SAME collectives: which parameters fired is agreed with a MAX all-reduce of a small
    bitmap over a host-side (gloo) control group in every ``wait()``; bucket ranges, the set of parameters cut
    out of their bucket and the "late" list are all derived from the agreed bitmap, never from local firing;
  * a parameter counts once per step (a second firing - a weight used twice - does not launch its bucket early);
    a gradient that arrives after its bucket has left is an error unless it is inside ``no_sync()`` - recorded by the
    hook, carried as a flag in the control exchange and raised by EVERY rank from ``wait()`` (a hook raising on one
    rank alone would leave its peers inside the next collective);
  * the control exchange is off the step's critical path once the firing set has settled: after ``STEADY_AFTER``
    consecutive steps with one and the same agreed set, ``wait()`` stops blocking on it - each step's bitmap goes out
    asynchronously and is looked at one step later (it completed long ago), the step itself assumes the settled set.
    A rank whose firing deviates in that mode keeps to the settled collective sequence (a gradient for a cut-out
    parameter is held back, not applied unreduced); every rank sees the deviation in the same ``wait()`` one step later,
    all return to the blocking exchange together, and the held-back gradient joins that step's reduction;
  * per-replica BatchNorm statistics are the design: ``average_bn_buffers`` averages the running
    statistics over the ranks, for the moment a state_dict is saved; dropout seeds differ per rank
    (``ops.DROPOUT_SEED_OFFSET``).
"""
import contextlib
import os

import torch
import torch.distributed as dist


def _forced():
    """WSDL_FORCE_DIST=1: run the whole data-parallel machinery (process group, broadcasts, bucketed all-reduces, control
    exchange) even with ONE rank - the only way to put the RCCL calls themselves on a one-GPU box."""
    return os.environ.get("WSDL_FORCE_DIST") == "1"


def _single(group=None):
    return dist.get_world_size(group) == 1 and not _forced()


def init_distributed(backend=None):
    """Initialise torch.distributed from the torchrun environment; returns (rank, local_rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or _forced()) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # WSDL_DIST_BACKEND=gloo rehearses the multi-process path on a box with fewer GPUs than ranks
            backend = os.environ.get("WSDL_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    if world > 1:
        from . import ops
        ops.DROPOUT_SEED_OFFSET[0] = rank * 0x9E3779B97F4A7C15 % (1 << 62)      # different masks on every replica
    return rank, local, world


_control_groups = {}


def control_group(group=None):
    """Host-side (gloo) twin of ``group`` for tiny control-plane exchanges that must not touch the GPU streams.  Always a
    communicator of its own: ranks reach the control exchange at different points of their collective sequence (one has
    launched a bucket from a hook, another has not yet), and collectives of ONE communicator must be issued in one order."""
    if not dist.is_initialized():
        return None
    key = id(group)
    if key not in _control_groups:
        ranks = dist.get_process_group_ranks(group) if group is not None else None
        _control_groups[key] = dist.new_group(ranks=ranks, backend="gloo")
    return _control_groups[key]


def sync_initial_state(optimizer, modules=(), group=None, src=0):
    """Broadcast the flat parameter buffer, the Adam moments / step count and every buffer of ``modules`` from ``src``."""
    if not dist.is_initialized() or _single(group):
        return
    for t in (optimizer.flat_param, optimizer.exp_avg, optimizer.exp_avg_sq):
        dist.broadcast(t, src=src, group=group)
    step = torch.tensor([optimizer.step_count], dtype=torch.int64)
    dist.broadcast(step, src=src, group=control_group(group))
    optimizer.step_count = int(step.item())
    if getattr(optimizer, "step_dev", None) is not None:
        optimizer.step_dev.fill_(optimizer.step_count)
    for m in modules:
        for b in m.buffers():
            dist.broadcast(b, src=src, group=group)
    from . import ops
    ops.bump_param_epoch()
    ops.bump_stats_epoch()


def average_bn_buffers(modules, group=None):
    """Replace every floating-point buffer (BatchNorm running_mean / running_var) by its mean over the ranks; call
    it before saving a state_dict, which would otherwise hold the saving rank's statistics only."""
    if not dist.is_initialized() or _single(group):
        return
    world = dist.get_world_size(group)
    for m in modules:
        if hasattr(m, "_join_aux"):
            m._join_aux()               # the train-mode aux head writes its running statistics on the side stream
        for b in m.buffers():
            if b.is_floating_point():
                dist.all_reduce(b, op=dist.ReduceOp.SUM, group=group)
                b.div_(world)
    from . import ops
    ops.bump_stats_epoch()


class GradBucketReducer:
    def __init__(self, optimizer, process_group=None, num_buckets=4, modules=(), sync_state=True, steady_after=None):
        self.opt = optimizer
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.active = dist.is_initialized() and (self.world > 1 or _forced())
        optimizer.grad_scale = 1.0 / self.world
        optimizer.pre_step_hook = self.wait
        optimizer._wsdl_reducer = self
        n = optimizer.numel
        nparams = len(optimizer.params)
        self._segmented = getattr(optimizer, "segments", None) is not None
        self._early = False
        if self._segmented:
            # the optimiser's segments ARE the buckets: small at the front of the buffer (the layers backward reaches
            # last), so the collective left exposed at the end of backward is the 1 MB one and not a quarter of the
            # gradients.  With optimizer.early_step a segment's Adam launch follows its all-reduce on the side stream
            # (measured on one rank over RCCL: 2 % slower than one launch after the join - opt-in).
            bounds = [lo for lo, _hi in optimizer.segments] + [n]
            self._early = bool(optimizer.early_step)
            optimizer.external_trigger = self.active and self._early
            if optimizer.external_trigger:
                optimizer.register_announce_hooks()
        else:
            num_buckets = max(1, min(num_buckets, nparams))
            # bucket boundaries on parameter boundaries, roughly equal sizes
            target = n / num_buckets
            bounds = [0]
            for off in optimizer.offsets:
                if off - bounds[-1] >= target and len(bounds) < num_buckets:
                    bounds.append(off)
            bounds.append(n)
        self.bounds = bounds
        self.bucket_of = []
        for off in optimizer.offsets:
            b = max(i for i in range(len(bounds) - 1) if bounds[i] <= off)
            self.bucket_of.append(b)
        self.bucket_size = [0] * (len(bounds) - 1)
        for b in self.bucket_of:
            self.bucket_size[b] += 1
        self._pending = []
        self._launched = [False] * len(self.bucket_size)
        self.early_launches = 0            # buckets of the last step that left before wait()
        # Parameters that receive no gradient (the aux head: no loss of the reference uses it) would hold their
        # bucket's collective back until the end of backward.  They are learnt from the previous step's AGREED
        # firing bitmap: unused parameters at either END of a bucket's range are cut out of its collective and no
        # longer waited for; a gradient that does turn up for one of them later is seen by every rank in the
        # agreed bitmap and reduced on its own at the join.
        self._fired = {}                   # parameter index -> source of its first announcement this step
        self._excluded = set()
        self._range = [(self.bounds[b], self.bounds[b + 1]) for b in range(len(self.bucket_size))]
        self._remaining = list(self.bucket_size)
        self._accumulating = False
        self._hooks = {}
        if steady_after is not None:
            self.STEADY_AFTER = int(steady_after)      # 0: always the blocking exchange (a late gradient is reduced in its own step)
        self._error = None                 # first hook error of this step (raised by every rank from wait())
        self._steady = False               # True: the control exchange is asynchronous, one step behind (module docstring)
        self._steady_set = None
        self._stable_steps = 0
        self._last_agreed = None
        self._ctl_work = self._ctl_buf = None
        self._held = {}                    # steady mode: gradients of cut-out parameters that fired here, held back one step
        self.control_exchanges_blocking = 0
        self.control_exchanges_async = 0
        # True while a rank verifies a freshly recorded launch plan: the verification's extra steps (an eager one on a probe
        # batch, the same batch through the replay) issue NO collective and take no part in the control exchange - ranks may
        # record at different moments (or not at all) without their collective sequences drifting apart
        self._local_only = False
        if self.active:
            self._control = control_group(process_group)
            self._bitmap = torch.zeros(nparams + 1, dtype=torch.uint8)          # last byte: error flag
            if sync_state:
                sync_initial_state(optimizer, modules, process_group)
            self._index = {id(p): i for i, p in enumerate(optimizer.params)}
            for i, p in enumerate(optimizer.params):
                h = self._make_hook(i)
                p.register_post_accumulate_grad_hook(lambda q, h=h: h(q, "autograd"))   # gradients arriving through autograd
            # gradients written directly by the HIP backward kernels into the flat buffer
            optimizer.grad_ready_hooks.append(lambda p: self._hooks[self._index[id(p)]](p, "direct"))

    @contextlib.contextmanager
    def no_sync(self):
        """Gradient accumulation: backward passes inside the context only accumulate into the flat buffer; the
        collectives of the step go out from the last backward (outside the context) or at ``wait()``."""
        self._accumulating = True
        try:
            yield
        finally:
            self._accumulating = False

    def _make_hook(self, i):
        b = self.bucket_of[i]

        def hook(_p, src):
            # Two sources can announce one gradient: "direct" (a HIP backward kernel has been enqueued that writes the
            # parameter's flat_grad slice) and "autograd" (the parameter's post-accumulate hook).  The engine runs the
            # post-accumulate hook of a parameter of a node in the graph even when that node's backward returned
            # None for it - which is what the direct path returns - so a direct firing is followed by a spurious
            # autograd one (measured: every one of the 182 trained parameters, every step).  The first announcement
            # of a step counts; the other source's echo is dropped.
            if self._accumulating:
                return
            # errors are RECORDED here and raised from wait() on every rank together (the flag travels in the control
            # exchange): an exception out of one rank's autograd hook would leave the peers inside the next collective
            if i in self._fired:
                if self._fired[i] == src and self._launched[b] and i not in self._excluded and self._error is None:
                    self._error = (f"a second gradient for parameter {i} {tuple(_p.shape)} arrived after its bucket's "
                                   "all-reduce had been launched (a second backward before step()?).  Wrap all but the "
                                   "last backward in reducer.no_sync().")
                return
            self._fired[i] = src
            if i in self._excluded:            # reduced on its own at the join (not part of the bucket's collective)
                return
            if self._launched[b]:
                if self._error is None:
                    self._error = f"gradient for parameter {i} of an already reduced bucket (see no_sync())"
                return
            self._remaining[b] -= 1
            self._launch_ready(early=True)
        self._hooks[i] = hook
        return hook

    def _launch_ready(self, early):
        """Buckets leave in ONE order on every rank - last bucket first, the order backward fills them - and a bucket
        only once every bucket behind it has left: ranks whose hooks fire in a different pattern (a data-dependent
        branch) still issue the same sequence of collectives, the stragglers from ``wait()``."""
        for b in range(len(self.bucket_size) - 1, -1, -1):
            if self._launched[b]:
                continue
            if early and self._remaining[b] > 0:
                return
            self._launch(b, early)

    def plan_ready(self):
        """May a training step under this reducer be recorded into / replayed from a launch plan (plan.PlannedTrainStep)?  Only
        once the firing set has settled (steady mode): then every step launches the same buckets from the same places and
        ``wait()`` finds nothing new - the collectives and ``wait()`` are the plan's host sections.  A rank that drops out
        of steady mode (a deviation seen in ``wait()``) goes back to eager steps; eager and replayed steps issue the SAME
        collectives in the same order, so ranks may even differ in which of the two they run."""
        if not self.active:
            return True
        return self._steady and not self._early and not self._held and self._error is None and not self._accumulating

    def _launch(self, b, early=False):
        lo, hi = self._range[b]
        if self.opt.flat_grad.is_cuda:
            from . import ops
            ops.flush_wgrad_reduces(self.opt.flat_grad.device)      # the bucket's weight gradients: slabs -> flat_grad, one launch
        if hi > lo and self.opt.flat_grad.is_cuda and not self._early:
            # behind everything both streams hold so far (recorded into a launch plan), then the collective itself as a host section
            from . import ops
            side = ops.side_stream(self.opt.flat_grad.device)
            ops.stream_wait(side, ops.raw_stream(self.opt.flat_grad.device))
            ops.host_section(self._launch_section, b, early)
            return
        self._launched[b] = True
        work = None
        if hi > lo:
            if early:
                self.early_launches += 1
            self._all_reduce(self.opt.flat_grad[lo:hi])
            work = self._pending[-1]
        if self._early:
            self.opt.step_segment(b, after=work)        # Adam on the bucket's parameters, behind its collective

    def _launch_section(self, b, early):
        """Bucket b's all-reduce from the side stream + the bookkeeping of ``_launch`` - the part of a bucket launch that a
        replayed plan repeats live (no gradient hook fires in a replay)."""
        self._launched[b] = True
        if early:
            self.early_launches += 1
        lo, hi = self._range[b]
        self._all_reduce(self.opt.flat_grad[lo:hi], ordered=True)

    def _all_reduce(self, view, ordered=False):
        if self._local_only:
            return          # verification of a launch plan (plan.PlannedTrainStep): the extra steps stay inside the rank
        if view.is_cuda:
            # The bucket's weight gradients are produced on the side stream, its BN / bias gradients on the main one.
            # Enqueue the collective behind BOTH from the side stream, so the main stream's dgrad chain never stalls.
            from . import ops
            ops.flush_wgrad_reduces(view.device)        # (nothing pending when _launch came first)
            side = ops.side_stream(view.device)
            if not ordered:
                ops.stream_wait(side, ops.raw_stream(view.device))
            with torch.cuda.stream(side):
                timed = getattr(self, "time_buckets", False)
                if timed:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                work = dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                if timed:
                    # measurement steps only (bench.py, after the timed region): the side stream waits for the collective
                    # right here, so that the second event marks its completion
                    work.wait()
                    e1.record()
                    self.__dict__.setdefault("_bucket_events", []).append((view.numel() * 4, e0, e1))
            self._pending.append(work)
            return
        self._pending.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    STEADY_AFTER = 3       # consecutive steps with one agreed firing set before the control exchange goes asynchronous

    def _local_bitmap(self):
        bm = self._bitmap
        bm.zero_()
        idx = list(self._fired) + list(self._held)
        if idx:
            bm[idx] = 1
        if self._error is not None:
            bm[-1] = 1
        return bm

    def _raise_together(self, flagged):
        if flagged:
            mine = self._error
            self._error = None
            raise RuntimeError("GradBucketReducer: " + (mine or "another rank reported a gradient-ordering error in this "
                                                        "step (see its message); every rank stops here together"))

    def _agree(self):
        """MAX over the ranks of the local firing bitmap (host-side control group): the one source of truth for what
        this step reduces beyond the buckets and for next step's ranges.  Blocking; also carries the error flag."""
        bm = self._local_bitmap()
        dist.all_reduce(bm, op=dist.ReduceOp.MAX, group=self._control)
        self.control_exchanges_blocking += 1
        self._raise_together(bool(bm[-1]))
        return set(torch.nonzero(bm[:-1]).flatten().tolist())

    def _join_side(self):
        opt = self.opt
        if opt.flat_grad.is_cuda:
            from . import ops
            ops.join_side_stream(opt.flat_grad.device)

    def wait(self):
        """Launch any bucket whose hooks did not all fire, reduce late gradients of excluded parameters, join, and
        learn which parameters to leave out next step - all from the bitmap every rank agrees on."""
        opt = self.opt
        if self.active and self._local_only:
            self._join_side()
            for b in range(len(self.bucket_size)):          # same ranges as last step: only re-arm the counters
                lo, hi = self._range[b]
                idx = [i for i, bb in enumerate(self.bucket_of) if bb == b and i not in self._excluded]
                self._remaining[b] = (idx[-1] - idx[0] + 1) if (idx and hi > lo) else 0
            self._pending = []
            self._fired = {}
            self._launched = [False] * len(self.bucket_size)
            self.early_launches = 0
            return
        if self.active:
            from . import ops as _ops
            if _ops.PLAN_REPLAYING[0]:
                # a replayed step: no gradient hook fired - the firing set is the settled one the plan was recorded under
                self._fired = dict(self._last_fired)
            self._last_fired = dict(self._fired)
            if self._steady and self._ctl_work is not None:
                # last step's exchange: launched a whole step ago, so this does not block in practice
                self._ctl_work.wait()
                prev, self._ctl_work = self._ctl_buf, None
                self._raise_together(bool(prev[-1]))
                if set(torch.nonzero(prev[:-1]).flatten().tolist()) != self._steady_set:
                    self._steady, self._stable_steps, self._last_agreed = False, 0, None     # every rank sees this in the same wait()
            if self._steady:
                fired = self._steady_set
                extra = sorted(i for i in self._fired if i in self._excluded)
                if extra:
                    self._join_side()                       # their weight gradients were written on the side stream
                    for i in extra:
                        off, n = opt.offsets[i], opt.params[i].numel()
                        g = opt.flat_grad[off:off + n]
                        self._held[i] = self._held[i] + g if i in self._held else g.clone()
                        g.zero_()                           # not applied unreduced: it joins the reduction one step later
                self._ctl_buf = self._local_bitmap().clone()
                self._ctl_work = dist.all_reduce(self._ctl_buf, op=dist.ReduceOp.MAX, group=self._control, async_op=True)
                self.control_exchanges_async += 1
            else:
                if self._held:
                    self._join_side()
                    for i, g in self._held.items():         # held back in steady mode: part of THIS step's reduction
                        off, n = opt.offsets[i], opt.params[i].numel()
                        opt.flat_grad[off:off + n] += g
                fired = self._agree()
                self._held = {}
            self._launch_ready(early=False)
            late = sorted(self._excluded & fired)
            if late and self._early:
                raise RuntimeError(
                    f"GradBucketReducer: parameters {late[:8]} received a gradient in this step but none in the previous "
                    "one; their segments have already been stepped with the unreduced values.  Set optimizer.early_step "
                    "= False and build the reducer on an optimiser without segments for models whose set of trained "
                    "parameters changes from step to step.")
            if late:
                self._join_side()                           # their weight gradients were written on the side stream
            for i in late:
                off, n = opt.offsets[i], opt.params[i].numel()
                self._all_reduce(opt.flat_grad[off:off + n])
            if not self._early:
                for w in self._pending:
                    w.wait()
                self._join_side()                           # the collectives were enqueued from the side stream
            # (early segment steps: every collective has been waited for on the side stream by the Adam launch behind it; the main
            # stream only waits for those launches - FlatAdam._finish_segments - not for the re-layouts queued after them)
            if not self._steady:
                # next step: per bucket, drop the unused parameters before the first / after the last used one
                self._excluded = set()
                for b in range(len(self.bucket_size)):
                    idx = [i for i, bb in enumerate(self.bucket_of) if bb == b]
                    used = [i for i in idx if i in fired]
                    if not used:
                        self._excluded.update(idx)
                        self._range[b] = (self.bounds[b], self.bounds[b])
                        self._remaining[b] = 0
                        continue
                    first, last = used[0], used[-1]
                    self._excluded.update(i for i in idx if i < first or i > last)
                    self._range[b] = (opt.offsets[first], opt.offsets[last] + opt.params[last].numel())
                    self._remaining[b] = last - first + 1       # unused parameters INSIDE the range are still waited for
                self._stable_steps = self._stable_steps + 1 if fired == self._last_agreed else 1
                self._last_agreed = fired
                if self._stable_steps >= self.STEADY_AFTER and self.STEADY_AFTER > 0:
                    self._steady, self._steady_set = True, fired
            else:
                for b in range(len(self.bucket_size)):          # same ranges as last step: only re-arm the counters
                    lo, hi = self._range[b]
                    idx = [i for i, bb in enumerate(self.bucket_of) if bb == b and i not in self._excluded]
                    self._remaining[b] = (idx[-1] - idx[0] + 1) if (idx and hi > lo) else 0
        self._pending = []
        self._fired = {}
        self._launched = [False] * len(self.bucket_size)
        self.last_early_launches, self.early_launches = self.early_launches, 0

    def bucket_times_ms(self):
        """Mean measured all-reduce time per collective size over the steps taken while ``time_buckets`` was set
        (``_all_reduce``): [{"bytes", "ms", "samples"}] in launch order of one step."""
        evs, self._bucket_events = self.__dict__.get("_bucket_events", []), []
        if not evs:
            return None
        torch.cuda.synchronize()
        order, acc = [], {}
        for nbytes, e0, e1 in evs:
            if nbytes not in acc:
                order.append(nbytes)
                acc[nbytes] = []
            acc[nbytes].append(e0.elapsed_time(e1))
        return [{"bytes": nb, "ms": round(sum(acc[nb]) / len(acc[nb]), 4), "samples": len(acc[nb])} for nb in order]

    def comm_budget(self, link_gbs=153.0):
        """Per bucket: bytes and the time its all-reduce needs on the node's xGMI mesh (each GPU has one link of
        ~``link_gbs`` GB/s to each of the other world-1): ``ring_us`` = 2 (N-1)/N x bytes through ONE link,
        ``direct_us`` = reduce-scatter + all-gather over all N-1 links at once = 2 x bytes / N per link.  The bucket left
        exposed at the end of backward is the first (smallest) one; every other bucket overlaps the remaining backward."""
        n = max(self.world, 1)
        out = []
        for b, (lo, hi) in enumerate(self._range):
            nbytes = 4 * (hi - lo)
            out.append({"bucket": b, "bytes": nbytes,
                        "ring_us": round(2 * (n - 1) / n * nbytes / (link_gbs * 1e3), 1) if n > 1 else 0.0,
                        "direct_us": round(2 * nbytes / n / (link_gbs * 1e3), 1) if n > 1 else 0.0})
        return out
