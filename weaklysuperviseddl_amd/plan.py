"""Launch plans: one host call per training iteration.

One iteration of the reference's training loop (TraditionalModel/AlternatingDirectionCutLoss.py:693-703 =
SegmentationModel.py:96-113) is ~520 kernel launches on three streams on this path.  Issued statement by statement from
Python - module call, autograd node, ctypes call, the entry point's tile choice - they cost the host 10-14 ms of an 18.7 ms
step; hipGraph replay on this ROCm costs 9.5 ms and runs slower on the GPU.  ``record`` runs a callable once in the
ordinary way while the library notes every launch it makes (function, grid, block, LDS, stream, argument values) and every
cross-stream dependency (``ops.stream_wait`` / ``ops.Event``) - ``include/wsdl_hip.h`` "launch plans", ``csrc/plan.hip`` -
and ``LaunchPlan.replay`` issues the same sequence again from one C loop.

``PlannedTrainStep`` is ``train_step`` on top of it: the first calls run eagerly, the next one is recorded, and - before a
single replay is trusted - VERIFIED: the model / optimiser state from before the recorded step is restored, the plan is
replayed on it, and parameters, both Adam moments, every module buffer, the device counters and the loss must come out
bit-identical to what the eager step left.  A step that contains anything a plan cannot see (kernels of the tensor library
in a user's ``extra_loss``, rocPRIM inside the Lovasz loss) fails that check, or the recording itself, and the step simply
stays eager (``PlannedTrainStep.disabled`` says why).

What a replay relies on: the recording keeps every tensor whose pointer went to the library alive, so no address is
ever handed to anybody else; whatever changes from step to step lives on the device (Adam's step number, the dropout call
counters, the amax slots' memset) or is an input copied into the plan's own input buffers; host-side twins (``step_count``,
BatchNorm's pending step counts, the cache epochs) are advanced by the wrapper.
"""
import ctypes as C
import os

import torch

from . import ops
from ._lib import lib, check, WsdlError


class PlanError(WsdlError):
    pass


class _Recording:
    __slots__ = ("keep", "sections")

    def __init__(self):
        self.keep = []
        self.sections = []          # host sections (ops.host_section): [(fn, args)] in recorded order


class LaunchPlan:
    """A recorded launch sequence (``record``).  ``keep``: everything the launches touch."""

    def __init__(self, handle, keep, sections=()):
        self.handle, self.keep, self.sections = handle, keep, list(sections)
        k, m, w, e, mk = (C.c_longlong(0) for _ in range(5))
        check(lib().wsdl_plan_stats(handle, C.byref(k), C.byref(m), C.byref(w), C.byref(e), C.byref(mk)))
        self.stats = {"kernels": k.value, "memsets": m.value, "stream_waits": w.value, "event_ops": e.value, "marks": mk.value}

    def replay(self):
        if not self.sections:
            check(lib().wsdl_plan_replay(self.handle))
            return
        # segments of launches with the recorded host work (collectives, waits on their handles) live in between
        ops.PLAN_REPLAYING[0] = True
        try:
            for k in range(len(self.sections) + 1):
                check(lib().wsdl_plan_replay_segment(self.handle, k))
                if k < len(self.sections):
                    fn, args = self.sections[k]
                    fn(*args)
        finally:
            ops.PLAN_REPLAYING[0] = False

    def replay_segment(self, k):
        check(lib().wsdl_plan_replay_segment(self.handle, int(k)))

    @property
    def segments(self):
        return self.stats["marks"] + 1

    def __del__(self):
        h, self.handle = getattr(self, "handle", None), None
        if h:
            try:
                lib().wsdl_plan_destroy(h)
            except Exception:           # interpreter shutdown
                pass


def recording():
    return ops.PLAN_REC[0] is not None


def record(fn, *args, **kwargs):
    """Run ``fn(*args, **kwargs)`` once, recording every launch of the library -> (LaunchPlan, fn's result).  Raises
    ``PlanError`` when the sequence cannot be replayed (the recorded call itself has run normally by then)."""
    if ops.PLAN_REC[0] is not None:
        raise PlanError("record: a plan is already being recorded")
    rec = _Recording()
    check(lib().wsdl_plan_begin())
    ops.PLAN_REC[0] = rec
    try:
        # The recording belongs to THIS host thread (csrc/plan.hip: thread_local).  The autograd engine runs the backward nodes
        # of a device on a worker thread of its own - their launches would not be recorded (and, were the recording process-wide,
        # a second thread's backward would be recorded into this plan).  With the engine's multithreading off for the recorded
        # call the whole backward pass runs on the calling thread: same nodes, same order.
        with torch.autograd.set_multithreading_enabled(False):
            out = fn(*args, **kwargs)
    except BaseException:
        ops.PLAN_REC[0] = None
        lib().wsdl_plan_abort()
        raise
    ops.PLAN_REC[0] = None
    h = C.c_void_p()
    if lib().wsdl_plan_end(C.byref(h)) != 0:
        err = PlanError(lib().wsdl_last_error().decode())
        err.result = out                # the call has run: its result is valid, only the plan is not
        raise err
    if len(rec.sections) != 0:
        n = C.c_longlong(0)
        check(lib().wsdl_plan_stats(h.value, None, None, None, None, C.byref(n)))
        if n.value != len(rec.sections):
            lib().wsdl_plan_destroy(h.value)
            err = PlanError("record: marks and host sections do not match")
            err.result = out
            raise err
    return LaunchPlan(h.value, rec.keep, rec.sections), out


PLAN_STEP = [os.environ.get("WSDL_PLAN_STEP", "1") != "0"]      # train_step replays a plan where it can (0: always eager)
PLAN_WARMUP = int(os.environ.get("WSDL_PLAN_WARMUP", "2"))      # eager calls before the recorded one
PLAN_DP = [os.environ.get("WSDL_PLAN_DP", "1") != "0"]           # replay under a GradBucketReducer too (0: data-parallel steps stay eager)


def _hook_tables(model):
    out = []
    for m in model.modules():
        for name in ("_forward_hooks", "_forward_pre_hooks", "_backward_hooks", "_backward_pre_hooks"):
            d = getattr(m, name, None)
            if d is not None:
                out.append(d)
    return out


MAX_PLANS = 2       # plans kept per training step (each holds a step's activations): the usual batch and one odd-sized one


class _Entry:
    __slots__ = ("plan", "s_images", "s_masks", "s_loss", "bn_delta", "sites")

    def __init__(self, plan, s_images, s_masks, s_loss, bn_delta, sites):
        self.plan, self.s_images, self.s_masks, self.s_loss, self.bn_delta, self.sites = plan, s_images, s_masks, s_loss, bn_delta, sites


class PlannedTrainStep:
    """``train_step(model, optimizer, images, masks, ...)`` as a verified plan replay (module docstring)."""

    def __init__(self, model, optimizer, eager, warmup=None):
        from . import nn as wnn
        self.model, self.opt, self.eager = model, optimizer, eager
        self.warmup = PLAN_WARMUP if warmup is None else int(warmup)
        self.calls = 0
        self.plan = None                # the plan of the most recent replay / recording (``entries`` holds one per key)
        self.key = None
        self.entries = {}               # key -> _Entry: a plan per (shapes, options) - e.g. an epoch's smaller last batch
        self.seen = {}                  # key -> eligible calls that found no plan for it
        self.disabled = None            # why this step stays eager, once it does
        self.replays = 0
        self.records = 0
        self._bns = [m for m in model.modules() if isinstance(m, wnn.BatchNorm2d)]
        self._convs = [m for m in model.modules() if isinstance(m, wnn.Conv2d)]
        self._dropouts = [m for m in model.modules() if isinstance(m, wnn.Dropout)]
        self._hooks = _hook_tables(model)
        self.loss_scalars = None        # host scalars of the step's loss objects (train_step sets it per call: ``host_scalars``)

    # -- what must be equal for a recorded plan to stand for the call: a plan freezes every HOST scalar that was a kernel
    # argument when it was recorded - BatchNorm's momentum / eps, a dropout's p, a loss object's weights and window sizes.  They
    # are all part of the key, so changing one records a new plan instead of being ignored; what changes every step belongs in
    # device memory (as Adam's hyper-parameters are), or the step falls back to eager ("the step's key keeps changing").
    def _key(self, images, masks):
        opt = self.opt
        red = getattr(opt, "_wsdl_reducer", None)
        # (lr, betas, eps, grad_scale are NOT part of the key: the Adam kernel reads them from device memory - optim.FlatAdam.hyper_dev)
        return (tuple(images.shape), images.dtype, tuple(masks.shape), images.device,
                None if red is None else (id(red), id(red._steady_set)),
                tuple((b.training, b.momentum, b.eps) for b in self._bns),
                tuple((d.training, float(d.p)) for d in self._dropouts), self.loss_scalars,
                tuple(p.requires_grad for p in opt.params), ops.LAYOUT_EPOCH[0],
                ops.CONV_ARITH[0], ops.OVERLAP_WGRAD[0], ops.WGRAD_AFTER_DGRAD[0], ops.BN_RELU_BITS[0], ops.IDENTITY_LINK[0], ops.ASPP_MULTI[0], ops.ASPP_GROUP_FWD[0],
                ops.raw_stream(images.device))

    def usable(self, images, masks):
        opt = self.opt
        red = getattr(opt, "_wsdl_reducer", None)
        if red is None:
            dp_ok = opt.pre_step_hook is None and not opt.grad_ready_hooks
        else:
            # data parallel: once the reducer's firing set has settled every step issues the same collectives at the same
            # places - they become host sections of the plan (ops.host_section)
            dp_ok = PLAN_DP[0] and red.plan_ready() and not getattr(red, "time_buckets", False)
        return (self.disabled is None and PLAN_STEP[0] and images.is_cuda and masks.is_cuda and self.model.training
                and torch.is_grad_enabled() and ops.PLAN_REC[0] is None and not ops.PROF_ON[0]
                and dp_ok and not getattr(opt, "time_tail", False)
                and not getattr(opt, "capture_mode", False) and not any(map(len, self._hooks))
                and not torch.cuda.is_current_stream_capturing())

    def _state(self):
        """Every device tensor a training step changes in place (what the verification compares and restores)."""
        opt = self.opt
        ts = [opt.flat_param, opt.exp_avg, opt.exp_avg_sq, opt.step_dev]
        ts += [b for b in self.model.buffers() if b.is_cuda and b.dim() > 0]
        ts += [m._counter for m in self._dropouts if m._counter is not None]
        return ts

    def _record(self, images, masks, key=None):
        # (the sentinel's host side stays quiet while a plan is recorded and verified: WSDL_RANGE_GUARD=auto changes library
        # options, which must not happen between the steps that are compared)
        self.opt._range_hold = True
        try:
            return self._record_held(images, masks, key)
        finally:
            self.opt._range_hold = False

    def _record_held(self, images, masks, key=None):
        dev = images.device
        opt = self.opt
        self.records += 1
        self.plan = None
        self.s_images = images.detach().clone()
        self.s_masks = masks.detach().to(torch.int64).clone()
        state = self._state()
        before = [t.clone() for t in state]
        pend0 = [b._pending_steps for b in self._bns]
        torch.cuda.synchronize(dev)
        ops.reset_amax_pool(dev)            # the first slot request inside the recording allocates a pool and memsets it there

        def restore(snap):
            for t, b in zip(state, snap):
                t.copy_(b)
            self._relayout()                # the layouts the step's forward reads belong to the restored parameters

        def reset_host(count, pend):
            opt.step_count = count
            for b, p in zip(self._bns, pend):
                b._pending_steps = p

        count0 = opt.step_count
        try:
            plan, loss = record(self.eager, self.s_images, self.s_masks)
        except PlanError as e:              # (the recorded call itself ran normally: a real training step was taken)
            self.disabled = f"recording failed: {e}"
            return e.result
        except Exception as e:              # the step itself failed while every tensor it touched was pinned by the recording
            # (out of memory first of all: a recording keeps all activations AND all normally transient gradients / workspaces
            # alive).  The step may be half done: back to the state before it, the recording's memory released, and the step
            # is taken eagerly - if it fails there too that error is the caller's.
            self.disabled = f"recording failed: {type(e).__name__}: {e}"
            del e
            torch.cuda.synchronize(dev)
            torch.cuda.empty_cache()
            restore(before)
            reset_host(count0, pend0)
            return self.eager(images, masks)
        finally:
            ops.reset_amax_pool(dev)        # eager code must not hand out the plan's slots
        self._bn_delta = [(b, b._pending_steps - p0) for b, p0 in zip(self._bns, pend0) if b._pending_steps != p0]
        # the state tensors a step touches may have grown (a dropout counter created by this very call)
        if len(self._state()) != len(state):
            return loss                     # (a dropout counter created by this very call) plan dropped: the next call records again
        # ---- verification (bit for bit), on a PROBE batch: with the recorded inputs a kernel the plan did not see would go
        # unnoticed - its output from the recorded run is still in memory and still right.  So: (1) back to the state before
        # the step, one EAGER step on a different batch -> reference; (2) back again, the same batch through the REPLAY ->
        # must equal the reference; (3) forward to the state the real step left.  Step (3) and the host counters are restored
        # WHATEVER happens in between (an out-of-memory error in the probe step - it runs on top of the recording's pinned
        # tensors, about twice an eager step's peak - must not cost the caller the real step it has already taken).
        after = [t.clone() for t in state]
        grad_after = opt.flat_grad.clone()
        loss_real = loss.detach().clone()
        host1 = (opt.step_count, [b._pending_steps for b in self._bns])

        probe_img = self.s_images * 0.75 + 0.1
        probe_masks = 1 - torch.clamp(self.s_masks, max=1)
        red = getattr(opt, "_wsdl_reducer", None)
        if red is not None:
            red._local_only = True          # data parallel: the two verification steps exchange nothing (dp.GradBucketReducer)
        bad, loss_same, failure = None, False, None
        try:
            restore(before)
            ref_loss = self.eager(probe_img, probe_masks).clone()
            ref = [t.clone() for t in state]
            restore(before)
            self.s_images.copy_(probe_img)
            self.s_masks.copy_(probe_masks)
            plan.replay()
            bad = [i for i, (t, a) in enumerate(zip(state, ref)) if not torch.equal(t, a)]
            loss_same = torch.equal(loss.detach(), ref_loss)
        except Exception as e:
            failure = f"{type(e).__name__}: {e}"
        finally:
            if red is not None:
                red._local_only = False
            if failure is not None:
                torch.cuda.synchronize(dev)
                ref = ref_loss = None       # (what the failed attempt still holds)
                torch.cuda.empty_cache()
            restore(after)
            reset_host(*host1)
            opt.flat_grad.copy_(grad_after)     # p.grad holds the REAL step's gradients again, not the probe batch's
        if failure is not None:
            self.disabled = "verification could not run (" + failure + "): the step stays eager"
            del plan
            return loss_real
        if bad or not loss_same:
            self.disabled = ("verification failed: the replayed step differs from the eager one ("
                             + (f"{len(bad)} of {len(state)} state tensors" if bad else "the loss value")
                             + ") - the step contains work a plan does not see (kernels of the tensor library in a custom loss?)")
            return loss_real
        self.plan, self.s_loss = plan, loss.detach()
        # the weight-layout buffers the plan's launches read and rewrite: (module cache, weight, the (fwd, dgrad) pair as recorded)
        sites = [(m.__dict__["_wsdl_cache"], m.weight, m.__dict__["_wsdl_cache"]["prep"]) for m in self._convs
                 if m.__dict__.get("_wsdl_cache", {}).get("prep") is not None and m.weight.requires_grad]
        ent = _Entry(plan, self.s_images, self.s_masks, self.s_loss, self._bn_delta, sites)
        if key is not None:
            while len(self.entries) >= MAX_PLANS:
                self.entries.pop(next(iter(self.entries)))
            self.entries[key] = ent
        self.key = key
        return loss_real                    # (``loss`` itself lives in the plan: the next replay overwrites it)

    def _relayout(self):
        """Weight layouts of the CURRENT parameters, in place (what the optimiser does after its step)."""
        if self.opt.post_step_hook is not None:
            self.opt.post_step_hook()

    def _buffers_intact(self, ent):
        """Do the modules still hold the layout buffers the plan's launches point at?  (A cache miss elsewhere - an eval forward
        after the layout options changed - allocates new ones: the plan would go on writing the old.)"""
        def same(cur, rec):
            return cur is not None and all((a is None) == (b is None) and (a is None or a.data_ptr() == b.data_ptr())
                                           for a, b in zip(cur, rec))
        return all(same(cache.get("prep"), prep) for cache, _w, prep in ent.sites)

    def _layouts_current(self, ent):
        """Do those buffers hold the layouts of the CURRENT parameters?  Yes after a replay or an eager step (both end with the
        re-layout); no after parameters were written some other way (load_state_dict): then they are re-laid out in place."""
        ep = ops.PARAM_EPOCH[0]
        return all(cache.get("prep_key") == (ep, w._version, w.data_ptr()) for cache, w, _prep in ent.sites)

    def __call__(self, images, masks):
        self.calls += 1
        if self.calls <= self.warmup or not self.usable(images, masks):
            return self.eager(images, masks)
        key = self._key(images, masks)
        ent = self.entries.get(key)
        if ent is not None and not self._buffers_intact(ent):
            self.entries.pop(key)
            ent = None
        if ent is None:
            n = self.seen[key] = self.seen.get(key, 0) + 1
            if self.entries and n < 2:
                return self.eager(images, masks)        # a second shape seen once (an epoch's last batch): eager until it recurs
            if self.records >= 6 and self.replays < 2 * self.records:
                self.disabled = ("the step's key keeps changing (shapes, learning rate or options differ from call to call): "
                                 "recording costs three steps each time")
                return self.eager(images, masks)
            if len(self.seen) > 64:
                self.seen.clear()
            return self._record(images, masks, key)
        if not self._layouts_current(ent):
            self._relayout()
        self.opt.sync_hyper()                   # a learning-rate schedule: five floats in device memory, the plan stays
        self.plan, self.key, self.s_images, self.s_masks, self.s_loss = ent.plan, key, ent.s_images, ent.s_masks, ent.s_loss
        if images.data_ptr() != ent.s_images.data_ptr():
            ent.s_images.copy_(images)
        if masks.data_ptr() != ent.s_masks.data_ptr():
            ent.s_masks.copy_(masks)
        ent.plan.replay()
        self.replays += 1
        # host-side twins of what the replayed kernels did
        self.opt.step_count += 1
        for b, d in ent.bn_delta:
            b._pending_steps += d
        ops.bump_param_epoch()
        ops.bump_stats_epoch()
        ep = ops.PARAM_EPOCH[0]
        for cache, w, _prep in ent.sites:           # the replay re-laid the weights out in place: the caches stay valid
            cache["prep_key"] = (ep, w._version, w.data_ptr())
        self.opt.range_poll()                   # the sentinel's host side (the check kernel itself is part of the plan)
        return ent.s_loss.clone()


_SCALARS = (bool, int, float, str, type(None))


def host_scalars(obj, _depth=0):
    """The host scalars a loss object would turn into kernel arguments - hashable.  Objects: their own (and, for torch modules,
    their sub-modules') int / float / bool / str attributes; functions: their defaults and the scalars in their closure cells
    (objects in cells one level deep).  Part of a plan's key: a loss weight ramped per epoch records a new plan instead of being
    frozen at the value of the recorded step.  A loss that wants a value to change WITHOUT a new plan keeps it in a device
    tensor."""
    if obj is None or _depth > 2:
        return None
    if isinstance(obj, _SCALARS):
        return obj
    if torch.is_tensor(obj):
        return ("tensor", obj.data_ptr(), tuple(obj.shape)) if obj.is_cuda else ("host-tensor", tuple(obj.flatten().tolist()[:16]))
    if isinstance(obj, (tuple, list)):
        return tuple(host_scalars(v, _depth + 1) for v in obj[:16])
    code = getattr(obj, "__code__", None)
    if code is not None:                    # a function / lambda
        cells = tuple(host_scalars(c.cell_contents, _depth + 1) for c in (obj.__closure__ or ()))
        return ("fn", tuple(host_scalars(d, _depth + 1) for d in (obj.__defaults__ or ())), cells)
    d = getattr(obj, "__dict__", None)
    if d is None:
        return None
    own = tuple(sorted((k, v) for k, v in d.items() if isinstance(v, _SCALARS) and not k.startswith("__")))
    if isinstance(obj, torch.nn.Module):
        own += tuple(host_scalars(m, _depth + 1) for m in obj.children())
    return (type(obj).__name__, own)


def loss_tag(obj):
    """Which PlannedTrainStep a loss object belongs to.  An object: itself (``id``).  A plain function or lambda: its CODE and
    the identities of what it captured - a ``lambda o, i: 0.1 * ncut(o, i)`` written inline in the training loop is a new
    function object on every call, but the same step (keyed on ``id`` it would never get past the warm-up calls and would
    churn PlannedTrainStep objects without a word)."""
    if obj is None:
        return None
    code = getattr(obj, "__code__", None)
    if code is None:
        return id(obj)
    # scalars in cells belong to the KEY (host_scalars), not to the tag; objects count by identity
    cells = tuple("scalar" if isinstance(c.cell_contents, _SCALARS) else id(c.cell_contents) for c in (obj.__closure__ or ()))
    return (code, id(getattr(obj, "__self__", None)), cells)


def planned_step_for(model, optimizer, eager, tag):
    """The PlannedTrainStep of (model, optimizer, tag) - kept on the optimizer; ``eager(images, masks)`` is the step."""
    table = optimizer.__dict__.setdefault("_wsdl_planned", {})
    key = (id(model), tag)
    st = table.get(key)
    if st is None:
        if len(table) >= 4:
            table.pop(next(iter(table)))
        st = table[key] = PlannedTrainStep(model, optimizer, eager)
    return st
