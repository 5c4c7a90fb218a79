"""FlatAdam - torch.optim.Adam semantics (defaults betas=(0.9,0.999), eps=1e-8, no weight decay;
reference TraditionalModel/SegmentationModel.py:91) as ONE kernel launch over flat buffers.

Parameters are re-homed into one contiguous fp32 buffer (``p.data`` become views) and so are their
gradients (``p.grad`` views of one flat buffer), which is also what the data-parallel gradient
all-reduce buckets (``dp.GradBucketReducer``) operate on: 39.6 M parameters = 158.5 MB per replica.

Segment steps (``enable_early_step``; used under data parallelism, opt-in for a single process - measured 0.4 %
slower there: six Adam launches instead of one, nothing left to hide).  The update is element-wise, so the flat buffer
can be stepped in pieces.
Backward fills it from the last layer to the first; the buffer is cut (on parameter boundaries) into segments that grow
geometrically from the FIRST layers - 1 MB, 5 MB, 25 MB, then at most 48 MB each - and a segment's Adam launch, the
amax of its convolution weights and their re-layout for the next step are enqueued on the side stream as soon as
its last expected gradient has been announced, behind the weight-gradient kernels that produce it.  What is left when
``step()`` is called is the small first segment: the stem and layer1.  Which parameters are "expected" is learnt from the previous step (the aux head of the
segmentation model never receives a gradient); a gradient that turns up for a parameter whose segment has already been
stepped raises.  The same segments are the data-parallel reducer's buckets: the all-reduce exposed at the end of
backward is the 1 MB one.
"""
import contextlib
import torch

from . import ops

_ALIGN = 64   # floats; keeps every parameter 256-byte aligned inside the flat buffer
import os as _os
# what FlatAdam does when the sentinel fires: "auto" (the default since round 6: switch the range guards on for the steps that
# follow - the reference's fp32 has no range floor, so the default must not keep computing below it) | "warn" (say so, change
# nothing) | "off"
RANGE_GUARD_DEFAULT = "auto"
RANGE_GUARD = [_os.environ.get("WSDL_RANGE_GUARD", RANGE_GUARD_DEFAULT)]
RANGE_GUARD_ACTIVE = [False]                                  # "auto" has switched the guards on in this process
RANGE_SENTINEL = [_os.environ.get("WSDL_RANGE_SENTINEL", "1") != "0"]      # FlatAdam.step() checks the step's tensors for regions below the fp16x2 arithmetic's safe range


class FlatAdam:
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, grad_scale=1.0):
        self.params = [p for p in params]
        if not self.params:
            raise ValueError("FlatAdam: empty parameter list")
        dev = self.params[0].device
        self.lr, self.betas, self.eps, self.grad_scale = lr, betas, eps, grad_scale
        self.step_count = 0
        self.step_dev = torch.zeros(1, device=dev, dtype=torch.int32) if dev.type == "cuda" else None   # device twin
        # lr, beta1, beta2, eps, grad_scale as the Adam kernel reads them (device floats: a schedule or a reducer changing one of
        # them changes memory, not a launch - a recorded launch plan of the step stays valid); sync_hyper() keeps them current
        self.hyper_dev = torch.zeros(5, device=dev, dtype=torch.float32) if dev.type == "cuda" else None
        self._hyper_host = None
        self.offsets, n = [], 0
        for p in self.params:
            self.offsets.append(n)
            n += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        self.numel = n
        self.flat_param = torch.zeros(n, device=dev, dtype=torch.float32)
        self.flat_grad = torch.zeros(n, device=dev, dtype=torch.float32)
        self.exp_avg = torch.zeros(n, device=dev, dtype=torch.float32)
        self.exp_avg_sq = torch.zeros(n, device=dev, dtype=torch.float32)
        with torch.no_grad():
            for p, off in zip(self.params, self.offsets):
                view = self.flat_param[off:off + p.numel()].view_as(p)
                view.copy_(p.data)
                p.data = view
                p.grad = self.flat_grad[off:off + p.numel()].view_as(p)
        self.segments = None           # early segment steps: [(lo, hi)] on parameter boundaries, None = one launch in step()
        self.pre_step_hook = None      # e.g. wait for the gradient all-reduce
        self.post_step_hook = None     # e.g. prefetch next step's weight layouts on the side stream
        # direct gradient delivery: the HIP backward kernels write each parameter's gradient straight into
        # its flat_grad slice (ops._sink_of); `_fresh` = not yet written since zero_grad (first write overwrites)
        self._fresh = {id(p): True for p in self.params}
        self.grad_ready_hooks = []     # callables(param): fired when a parameter's gradient has been written
        for p in self.params:
            p._wsdl_grad_sink = self

    # ---------------------------------------------------------------------------------- early segment steps
    def enable_early_step(self, segment_hook=None, first=1 << 18, growth=5, cap=12_000_000):
        """Cut the flat buffer into segments and step each as soon as its gradients are complete (module docstring).
        ``segment_hook(k, param_indices, use_events, adam_event)`` runs right after segment k's Adam launch: the place to
        re-lay-out that segment's convolution weights (behind ``adam_event``, on a stream of its own)."""
        if not self.flat_param.is_cuda:
            return self
        bounds, target = [0], first
        for off in self.offsets[1:]:
            if off - bounds[-1] >= target:
                bounds.append(off)
                target = min(target * growth, cap)
        if len(bounds) > 1 and self.numel - bounds[-1] < first:
            bounds.pop()                               # no crumb at the end
        bounds.append(self.numel)
        self.segments = [(bounds[i], bounds[i + 1]) for i in range(len(bounds) - 1)]
        self.seg_of = []
        for off in self.offsets:
            self.seg_of.append(max(k for k, (lo, _hi) in enumerate(self.segments) if lo <= off))
        self.seg_params = [[i for i, k in enumerate(self.seg_of) if k == kk] for kk in range(len(self.segments))]
        self.segment_hook = segment_hook
        self.early_step = False         # True: step segments as backward completes them (False: one launch in step(),
                                        # unless a data-parallel reducer drives the segments - external_trigger)
        self.external_trigger = False   # True: someone else (the DP reducer) decides when a segment is complete
        self.capture_mode = False       # inside hipGraph capture: no cross-replay events
        self._index = {id(p): i for i, p in enumerate(self.params)}
        self._expected = None           # parameter indices that fired in the previous step
        self._fired = set()
        self._stepped = [False] * len(self.segments)
        self._remaining = None
        self._accumulating = False
        self._adam_event = None
        self._prep_done = [None] * len(self.segments)     # per segment: event behind its last re-layout
        self._seg_events = [None] * len(self.segments)    # per segment: event behind its Adam launch
        self._hooked = False
        return self

    def register_announce_hooks(self):
        """Gradient announcements (needed only when segments are stepped before ``step()``: 364 Python hook calls per step
        of the segmentation model cost the host ~2.5 ms of the 16 it needs to enqueue a step)."""
        if self.segments is None or self._hooked:
            return
        self._hooked = True
        for i, p in enumerate(self.params):
            p.register_post_accumulate_grad_hook(lambda q, i=i: self._announce(i))
        self.grad_ready_hooks.append(lambda p: self._announce(self._index[id(p)]))

    @contextlib.contextmanager
    def accumulate(self):
        """Backward passes inside the context only accumulate gradients (no early segment step)."""
        self._accumulating = True
        try:
            yield
        finally:
            self._accumulating = False

    def _announce(self, i):
        if i in self._fired:
            return
        self._fired.add(i)
        k = self.seg_of[i]
        if self._stepped[k]:
            raise RuntimeError(
                f"FlatAdam: a gradient for parameter {i} {tuple(self.params[i].shape)} arrived after its segment had been "
                "stepped (a parameter that received no gradient in the previous step, or a second backward before "
                "step()).  Wrap extra backward passes in optimizer.accumulate() or set optimizer.early_step = False.")
        if self.external_trigger or not self.early_step or self.capture_mode or self._accumulating or self._expected is None:
            return
        if i in self._expected:
            self._remaining[k] -= 1
        # segments leave in backward order, a segment only once every later one has left
        for kk in range(len(self.segments) - 1, -1, -1):
            if self._stepped[kk]:
                continue
            if self._remaining[kk] > 0:
                break
            self.step_segment(kk)

    def step_segment(self, k, after=None):
        """Adam on segment k, enqueued on the side stream behind the main stream's work so far (BatchNorm / bias
        gradients, the input-gradient kernels that still read this segment's weight layouts) and behind ``after``
        (a collective's work handle)."""
        dev = self.flat_param.device
        lo, hi = self.segments[k]
        ops.flush_wgrad_reduces(dev)            # the segment's weight gradients may still be un-reduced slabs
        side = ops.side_stream(dev)
        ops.stream_wait(side, ops.raw_stream(dev))
        if not self.capture_mode and self._prep_done[k] is not None:
            self._prep_done[k].wait(side)              # last step's re-layouts of THIS segment read the parameters overwritten now
        ev = None
        with torch.cuda.stream(side):
            if after is not None:
                after.wait()
            if not any(self._stepped):
                self.step_count += 1
                ops.add_int(self.step_dev, 1)
            ops.adam_step_flat(self.flat_param[lo:hi], self.flat_grad[lo:hi], self.exp_avg[lo:hi], self.exp_avg_sq[lo:hi],
                               self.lr, self.betas[0], self.betas[1], self.eps, self.step_count, self.grad_scale,
                               step_dev=self.step_dev, hyper_dev=self.hyper_dev)
            if not self.capture_mode:
                ev = self._seg_events[k]               # one event per segment, re-recorded step after step
                if ev is None:
                    ev = self._seg_events[k] = ops.Event()
                self._adam_event = ev
                ev.record(side)
        self._stepped[k] = True
        if self.segment_hook is not None:
            self._prep_done[k] = self.segment_hook(k, self.seg_params[k], not self.capture_mode, ev)

    def _finish_segments(self):
        dev = self.flat_param.device
        for k in range(len(self.segments) - 1, -1, -1):
            if not self._stepped[k]:
                self.step_segment(k)
        if self.capture_mode or self._adam_event is None:
            ops.stream_wait(ops.raw_stream(dev), ops.side_stream(dev))
        else:
            self._adam_event.wait(ops.raw_stream(dev))   # the parameters, not the re-layouts behind them on the side stream
        self._expected = set(self._fired)
        self._fired = set()
        self._stepped = [False] * len(self.segments)
        self._remaining = [sum(1 for i in idx if i in self._expected) for idx in self.seg_params]
        self._backward_pending = False

    def sync_hyper(self):
        """Copy (lr, beta1, beta2, eps, grad_scale) to the device when one of them changed since the last call.

        Never inside a stream capture: the copy is a pageable host-to-device transfer (illegal in a hipGraph capture, and a
        captured copy would freeze the values anyway).  ``GraphedTrainStep`` calls this BEFORE it captures and before every
        replay, so the captured Adam node reads current values from ``hyper_dev``; a change that turns up while a capture is
        running is an error, not something to skip silently."""
        cur = (float(self.lr), float(self.betas[0]), float(self.betas[1]), float(self.eps), float(self.grad_scale))
        if self.hyper_dev is not None and cur != self._hyper_host:
            if self.hyper_dev.is_cuda and torch.cuda.is_current_stream_capturing():
                raise RuntimeError("FlatAdam.sync_hyper: lr / betas / eps / grad_scale changed inside a stream capture; change "
                                   "them between steps (GraphedTrainStep copies them to the device before each replay)")
            self.hyper_dev.copy_(torch.tensor(cur, dtype=torch.float32))
            self._hyper_host = cur

    def take_fresh(self, p):
        f = self._fresh.get(id(p), False)
        self._fresh[id(p)] = False
        return f

    def grad_ready(self, p):
        self._dirty = True              # the side stream may be writing flat_grad: zero_grad must join before its memset
        for h in self.grad_ready_hooks:
            h(p)

    def zero_grad(self, set_to_none=False):
        self.sync_hyper()               # (segments may be stepped from backward hooks, before step() is reached)
        if self.flat_grad.is_cuda:
            ops.flush_wgrad_reduces(self.flat_grad.device)   # reductions still pending would land in the zeroed buffer
        if self.flat_grad.is_cuda and getattr(self, "_dirty", True):
            ops.join_side_stream(self.flat_grad.device)      # a backward without a step may still be writing
        if self.flat_grad.is_cuda:
            ops.memset_zero(self.flat_grad)         # a launch of the library: part of a recorded plan
        else:
            self.flat_grad.zero_()
        for k in self._fresh:
            self._fresh[k] = True
        for p, off in zip(self.params, self.offsets):
            if p.grad is None or p.grad.data_ptr() != self.flat_grad.data_ptr() + 4 * off:
                p.grad = self.flat_grad[off:off + p.numel()].view_as(p)

    def tail_ms(self):
        """Mean main-stream time of the optimiser tail over the steps taken while ``time_tail`` was set: from the entry of
        ``step()`` (backward's last main-stream kernel is enqueued) to its exit (the parameters are written: the next forward
        may start).  Single process: the join with the weight-gradient stream + Adam.  Under ``GradBucketReducer``: also what
        the main stream waits for the gradient collectives - the EXPOSED communication is this figure minus the
        single-process one."""
        pairs, self._tail_events = getattr(self, "_tail_events", []), []
        if not pairs:
            return None
        torch.cuda.synchronize(self.flat_param.device)
        return sum(a.elapsed_time(b) for a, b in pairs) / len(pairs)

    def step(self):
        if getattr(self, "time_tail", False) and self.flat_param.is_cuda and not self.capture_mode:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            try:
                return self._step()
            finally:
                e1.record()
                self.__dict__.setdefault("_tail_events", []).append((e0, e1))
        return self._step()

    def range_poll(self):
        """Host side of the range sentinel (no synchronisation: the check kernels of the step before write pinned host memory):
        warn once when a tensor of the last step spanned more than the fp16x2 arithmetic's safe range - and, with
        ``WSDL_RANGE_GUARD=auto``, switch both range guards on for the steps that follow (weight layouts and launch plans
        re-form by themselves: ops.set_option bumps their epochs).  Called by the eager step and after every plan replay."""
        if getattr(self, "_range_warned", False) or getattr(self, "_range_hold", False) or RANGE_GUARD[0] == "off" or \
                not (RANGE_SENTINEL[0] and ops.CONV_ARITH[0] == 1):
            return
        st = ops.range_status(self.flat_param.device)
        if not st["exceeded"]:
            return
        self._range_warned = True
        import warnings
        what = (f"weaklysuperviseddl_amd: {st['pairs_over_limit']} activation / gradient tensors of the last step span more than "
                f"2^{st['limit_log2']} between their largest and their smallest region (worst 2^{st['worst_log2']:.0f}): the default "
                "convolution arithmetic (fp16x2, one scale per tensor) computes the small regions to fewer than 13 bits there - ")
        if RANGE_GUARD[0] == "auto":
            ops.set_option("conv_arith", 2)
            ops.set_option("wgrad_chan_scale", 1)
            RANGE_GUARD_ACTIVE[0] = True
            warnings.warn(what + "WSDL_RANGE_GUARD=auto: the range guards are on from the next step (conv_arith = 2, wgrad_chan_scale = 1; "
                          "about 10 % slower)")
        else:
            warnings.warn(what + "select the range guards: ops.set_option('conv_arith', 2) (forward / input gradient) and "
                          "ops.set_option('wgrad_chan_scale', 1) (weight gradient), or WSDL_RANGE_GUARD=auto")

    def _sentinel(self):
        """Range sentinel of the fp16x2 arithmetic (every exit of a step, the segmented / data-parallel one too): what the
        PREVIOUS step's tensors spanned, then this step's check - behind the join with the side stream, whose pools it reads."""
        if self.flat_param.is_cuda and RANGE_SENTINEL[0] and ops.CONV_ARITH[0] == 1 and not self.capture_mode_on():
            self.range_poll()
            ops.range_check(self.flat_param.device)

    def capture_mode_on(self):
        return bool(getattr(self, "capture_mode", False))

    def _step(self):
        self.sync_hyper()
        if self.flat_param.is_cuda:
            ops.flush_wgrad_reduces(self.flat_param.device)     # (normally done: the end of the backward pass flushed them)
        if self.pre_step_hook is not None:
            # (the data-parallel reducer's wait(): collectives' work handles, control-plane exchange - host work that a launch
            # plan repeats live at this place)
            ops.host_section(self.pre_step_hook)
        if self.segments is not None:
            if self.external_trigger or self.early_step or any(self._stepped):
                self._finish_segments()         # (the segment hook has re-laid-out the weights)
                ops.bump_param_epoch()
                self._dirty = False             # the main stream waits for the Adam launches, which follow every gradient
                self._sentinel()
                return
            self._fired = set()                 # one launch over the whole buffer, below
        if self.flat_param.is_cuda:
            ops.join_side_stream(self.flat_param.device)     # weight gradients are produced on the side stream
        self.step_count += 1
        ops.bump_param_epoch()          # parameters change behind torch's version counters
        if self.flat_param.is_cuda:
            # the kernel reads the step number from the device: a captured (hipGraph) step replays correctly
            ops.add_int(self.step_dev, 1)
            ops.adam_step_flat(self.flat_param, self.flat_grad, self.exp_avg, self.exp_avg_sq, self.lr,
                               self.betas[0], self.betas[1], self.eps, self.step_count, self.grad_scale,
                               step_dev=self.step_dev, hyper_dev=self.hyper_dev)
        else:
            raise ops.WsdlError("FlatAdam.step: parameters are not on the device; there is no CPU fallback")
        self._dirty = False                     # (joined above)
        self._sentinel()
        if self.post_step_hook is not None:
            self.post_step_hook()
