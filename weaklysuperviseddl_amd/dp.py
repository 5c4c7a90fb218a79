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
post-accumulate-grad hook of the last parameter it is waiting for and overlaps the remaining backward;
``wait()`` (called from ``FlatAdam.step``) only makes the compute stream wait on those collectives.
"""
import os

import torch
import torch.distributed as dist


def init_distributed(backend=None):
    """Initialise torch.distributed from the torchrun environment; returns (rank, local_rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # WSDL_DIST_BACKEND=gloo rehearses the multi-process path on a box with fewer GPUs than ranks
            backend = os.environ.get("WSDL_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


class GradBucketReducer:
    def __init__(self, optimizer, process_group=None, num_buckets=4):
        self.opt = optimizer
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        optimizer.grad_scale = 1.0 / self.world
        optimizer.pre_step_hook = self.wait
        n = optimizer.numel
        num_buckets = max(1, min(num_buckets, len(optimizer.params)))
        # bucket boundaries on parameter boundaries, roughly equal sizes
        target = n / num_buckets
        bounds, acc = [0], 0
        for p, off in zip(optimizer.params, optimizer.offsets):
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
        # Parameters that receive no gradient (the aux head: no loss of the reference uses it) would hold their
        # bucket's collective back until the end of backward.  They are learnt from the previous step: unused
        # parameters at either END of a bucket's range are cut out of its collective (so a gradient that does turn up
        # for one of them later cannot race with it; it is reduced on its own at the join) and no longer waited for.
        self._fired = set()
        self._excluded = set()
        self._range = [(self.bounds[b], self.bounds[b + 1]) for b in range(len(self.bucket_size))]
        self._remaining = list(self.bucket_size)
        if self.world > 1:
            self._index = {id(p): i for i, p in enumerate(optimizer.params)}
            for i, p in enumerate(optimizer.params):
                p.register_post_accumulate_grad_hook(self._make_hook(i))     # gradients arriving through autograd
            # gradients written directly by the HIP backward kernels (no AccumulateGrad node runs for them)
            optimizer.grad_ready_hooks.append(lambda p: self._hooks[self._index[id(p)]](p))

    def _make_hook(self, i):
        b = self.bucket_of[i]

        def hook(_p):
            self._fired.add(i)
            if i in self._excluded:            # reduced on its own at the join (not part of the bucket's collective)
                return
            self._remaining[b] -= 1
            if self._remaining[b] == 0:
                self._launch(b)
        if not hasattr(self, "_hooks"):
            self._hooks = {}
        self._hooks[i] = hook
        return hook

    def _launch(self, b):
        if self._launched[b]:
            return
        self._launched[b] = True
        lo, hi = self._range[b]
        if hi <= lo:
            return
        view = self.opt.flat_grad[lo:hi]
        if view.is_cuda:
            # The bucket's weight gradients are produced on the side stream, its BN / bias gradients on the main one.
            # Enqueue the collective behind BOTH from the side stream, so the main stream's dgrad chain never stalls.
            from . import ops
            main, side = torch.cuda.current_stream(view.device), ops.side_stream(view.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                work = dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self._pending.append(work)
            return
        self._pending.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def wait(self):
        """Launch any bucket whose hooks did not all fire, reduce late gradients of excluded parameters, join, and
        learn which parameters to leave out next step."""
        opt = self.opt
        if self.world > 1:
            for b in range(len(self.bucket_size)):
                self._launch(b)
            late = sorted(self._excluded & self._fired)
            if late and opt.flat_grad.is_cuda:
                from . import ops
                ops.join_side_stream(opt.flat_grad.device)      # their weight gradients were written on the side stream
            for i in late:
                off, n = opt.offsets[i], opt.params[i].numel()
                self._pending.append(dist.all_reduce(opt.flat_grad[off:off + n], op=dist.ReduceOp.SUM, group=self.group,
                                                     async_op=True))
            for w in self._pending:
                w.wait()
            # next step: per bucket, drop the unused parameters before the first / after the last used one
            self._excluded = set()
            for b in range(len(self.bucket_size)):
                idx = [i for i, bb in enumerate(self.bucket_of) if bb == b]
                used = [i for i in idx if i in self._fired]
                if not used:
                    self._excluded.update(idx)
                    self._range[b] = (self.bounds[b], self.bounds[b])
                    self._remaining[b] = 0
                    continue
                first, last = used[0], used[-1]
                self._excluded.update(i for i in idx if i < first or i > last)
                self._range[b] = (opt.offsets[first], opt.offsets[last] + opt.params[last].numel())
                self._remaining[b] = last - first + 1       # unused parameters INSIDE the range are still waited for
        self._pending = []
        self._fired = set()
        self._launched = [False] * len(self.bucket_size)
