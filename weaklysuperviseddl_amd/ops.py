"""Host-side glue between PyTorch tensors (device memory, streams, autograd bookkeeping) and the HIP
kernels of libwsdl_hip.so.  Every arithmetic step runs in the library; torch here only allocates
buffers and orders calls on the current stream.  No CPU fallback: non-device tensors raise.
"""
import ctypes as C

import os

import threading

import torch

from ._lib import lib, check, WsdlError

_vp = C.c_void_p


# torch.cuda.current_stream() builds a Stream object (and resolves the device twice) per call: ~6.5 us, ~350 times per
# training step = 2.3 ms of the host's ~13 ms (tools/host_profile.py).  The raw handle is one C call.
_raw_stream = torch._C._cuda_getCurrentRawStream
_cur_device = torch._C._cuda_getDevice


def raw_stream(device=None):
    """hipStream_t of the current stream of ``device`` (default: the current device) as an int."""
    idx = device.index if (device is not None and device.index is not None) else _cur_device()
    return _raw_stream(idx)


def _stream():
    return _raw_stream(_cur_device())          # an int: the bound functions declare c_void_p (argtypes), ctypes converts


# ---- launch plans (plan.py; include/wsdl_hip.h "launch plans") ---------------------------------------------------------
# While a plan is being recorded, PLAN_REC[0] is the recording.  Everything the recorded launches touch must outlive the
# plan AT ITS ADDRESS: every tensor whose pointer is handed to the library meanwhile is kept by the recording (``_p``), so
# the caching allocator can never give its block to anybody else - no private memory pool needed.
# Both are PER HOST THREAD (as the C side's recording is: csrc/plan.hip): a second thread that uses the library while this
# one records - a loader worker, a second model - launches normally, pins nothing into this thread's recording and may
# record a plan of its own.
class _ThreadSlot(threading.local):
    """``slot[0]`` with one value per host thread."""

    def __init__(self, default):            # (threading.local calls this once per thread that touches the slot)
        self.v = default

    def __getitem__(self, i):
        return self.v

    def __setitem__(self, i, value):
        self.v = value


PLAN_REC = _ThreadSlot(None)


PLAN_REPLAYING = _ThreadSlot(False)     # a plan with host sections is being replayed (the sections run live, in their places)


def host_section(fn, *args):
    """Run ``fn(*args)`` - host work that must happen at THIS place of the launch sequence on every iteration and that a
    plan cannot hold: a collective of torch.distributed, a wait on its work handle, control-plane exchanges.  Outside a
    recording it is a plain call.  While a plan is being recorded the plan is cut here (``wsdl_plan_mark``), the
    recording pauses for the call, and a replay calls ``fn(*args)`` again between the two segments."""
    rec = PLAN_REC[0]
    if rec is None:
        return fn(*args)
    check(lib().wsdl_plan_mark(len(rec.sections)))
    check(lib().wsdl_plan_pause())
    PLAN_REC[0] = None
    try:
        return fn(*args)
    finally:
        PLAN_REC[0] = rec
        check(lib().wsdl_plan_resume())
        rec.sections.append((fn, args))


def _p(t):
    if t is None:
        return None
    rec = PLAN_REC[0]
    if rec is not None:
        rec.keep.append(t)
    return t.data_ptr()                                 # int -> c_void_p by the declared argtypes (no object per argument)


def _h(stream):
    """Raw hipStream_t (an int) of a torch stream object / of a raw handle."""
    return stream if isinstance(stream, int) else stream.cuda_stream


def stream_wait(waiter, waited):
    """``waiter`` waits for everything enqueued on ``waited`` so far (torch stream objects or raw handles).  Through the
    library (one event record + one stream wait), so that a plan being recorded sees the dependency."""
    check(lib().wsdl_stream_wait_stream(_h(waiter), _h(waited)))


class Event:
    """An event of the library (no timing).  ``record`` / ``wait`` default to torch's current stream; both are part of a plan
    being recorded (which then keeps the event alive)."""
    __slots__ = ("h", "__weakref__")

    def __init__(self):
        h = C.c_void_p()
        check(lib().wsdl_event_create(C.byref(h)))
        self.h = h.value

    def record(self, stream=None):
        if PLAN_REC[0] is not None:
            PLAN_REC[0].keep.append(self)
        check(lib().wsdl_event_record(self.h, _stream() if stream is None else _h(stream)))

    def wait(self, stream=None):
        if PLAN_REC[0] is not None:
            PLAN_REC[0].keep.append(self)
        check(lib().wsdl_stream_wait_event(_stream() if stream is None else _h(stream), self.h))

    def __del__(self):
        h, self.h = self.h, None
        if h:
            try:
                lib().wsdl_event_destroy(h)
            except Exception:       # interpreter shutdown
                pass


def cross_stream_use(t, stream):
    """``t`` lives on another stream's allocator pool and is used by work enqueued on ``stream`` (a torch stream object):
    keep the caching allocator from recycling it before that work has run."""
    t.record_stream(stream)
    if PLAN_REC[0] is not None:
        PLAN_REC[0].keep.append(t)


def memset_zero(t):
    """t.zero_() as a stream-ordered memset of the library (seen by a plan; ``t`` dense)."""
    if not t.is_contiguous():
        raise WsdlError("memset_zero: tensor is not dense")
    if torch.cuda.is_current_stream_capturing():
        return t.zero_()        # inside a hipGraph capture: the tensor library's fill kernel, as the captured graphs always had
    check(lib().wsdl_memset_async(_p(t), 0, t.numel() * t.element_size(), _stream()))
    return t


def add_int(t, delta=1):
    """t += delta for a one-element device int32 / int64 (Adam's step number, a dropout call counter) - a launch of the
    library instead of torch's ``add_`` (a plan records launches of this library only)."""
    if t.numel() != 1 or t.dtype not in (torch.int32, torch.int64):
        raise WsdlError("add_int: a one-element int32 / int64 device tensor")
    check(lib().wsdl_add_int(_p(t), int(t.dtype == torch.int64), int(delta), _stream()))
    return t


def clamp_max_labels(labels, hi=1):
    """torch.clamp(labels, max=hi) for int64 device labels (reference SegmentationModel.py:100); other dtypes / host
    tensors take torch's own clamp."""
    if not labels.is_cuda or labels.dtype != torch.int64:
        return torch.clamp(labels, max=hi)
    labels = labels.contiguous()
    out = torch.empty_like(labels)
    if labels.numel():
        check(lib().wsdl_clamp_max_i64(_p(labels), _p(out), labels.numel(), int(hi), _stream()))
    return out


def _req(t, name="tensor", dtype=torch.float32):
    if not t.is_cuda:
        raise WsdlError(f"{name}: the HIP path needs a device tensor (got {t.device}); there is no CPU fallback")
    if t.dtype != dtype:
        raise WsdlError(f"{name}: expected {dtype}, got {t.dtype}")
    return t


def _dense(t, name="tensor"):
    _req(t, name)
    return t if t.is_contiguous() else t.contiguous()


def _planes(t, name="tensor"):
    """(tensor, batch_stride): accept NCHW tensors whose images are dense but batch stride is larger
    (channel slices of a concatenated tensor) without copying; anything else is made contiguous."""
    _req(t, name)
    B, Cc, H, W = t.shape
    if t.is_contiguous():
        return t, Cc * H * W
    st = t.stride()
    if st[3] == 1 and st[2] == W and st[1] == H * W and st[0] >= Cc * H * W:
        return t, st[0]
    t = t.contiguous()
    return t, Cc * H * W


_ws_cache = {}

# Parameters / BN statistics are also written by raw-pointer kernels (Adam step, train-mode running statistics),
# which torch's tensor version counters cannot see.  Every such writer bumps PARAM_EPOCH; eval-mode caches of
# derived tensors (re-laid-out weights, folded BN scale/shift) are keyed on (PARAM_EPOCH, tensor._version, data_ptr).
PARAM_EPOCH = [0]      # parameters rewritten by the Adam kernel
STATS_EPOCH = [0]      # BN running statistics rewritten by a train-mode forward


def bump_param_epoch():
    PARAM_EPOCH[0] += 1


def bump_stats_epoch():
    STATS_EPOCH[0] += 1


def _cache_key(*tensors):
    return (PARAM_EPOCH[0], STATS_EPOCH[0]) + tuple((t._version, t.data_ptr()) for t in tensors)


def _weight_key(w):
    return (PARAM_EPOCH[0], w._version, w.data_ptr())


def _cached_prep(cache, weight, need_dx):
    """(wt_fwd, wt_dgrad) from the module cache when the weights have not changed since they were laid out (eval
    mode, or prefetched on the side stream right after the optimiser step); otherwise lay them out now."""
    if cache is not None and cache.get("prep_key") == _weight_key(weight) and (cache["prep"][1] is not None or not need_dx):
        ev = cache.get("prep_event")
        if ev is not None:
            ev.wait()
        if PLAN_REC[0] is not None:
            PLAN_REC[0].keep.append(cache["prep"])
        return cache["prep"]
    wf, wd = prep_weights(weight, True, need_dx)
    if cache is not None:
        cache["prep_key"], cache["prep"], cache["prep_event"] = _weight_key(weight), (wf, wd), None
    return wf, wd


_PREFETCH_HEAD = int(os.environ.get("WSDL_PREFETCH_HEAD", "12"))    # convolutions of the first re-layout batch after an optimiser step (stem + layer1)
LAYOUT_EPOCH = [0]     # bumped by options that change what a layout buffer holds / how large it is


_prep_streams = {}


def _norm_device(device):
    """torch.device with an explicit index: the stream tables and the census must agree on the key (an index-less
    ``torch.device('cuda')`` used to create streams the census did not count)."""
    device = device if isinstance(device, torch.device) else torch.device(device)
    if device.type == "cuda" and device.index is None:
        device = torch.device("cuda", _cur_device())
    return device


def prep_stream(device):
    """Third stream: re-layouts that follow an early segment step must not sit in the side stream in front of the weight
    gradients (and the Adam launches) still to come."""
    if device.index is None:
        device = _norm_device(device)
    st = _prep_streams.get(device)
    if st is None:
        st = _prep_streams[device] = torch.cuda.Stream(device=device)
    return st


def prefetch_weight_layouts(convs, use_events=True, epoch_ahead=0, pingpong=False, after=None, _head=True):
    """Lay out next step's weights on the side stream (called right after the optimiser step of these weights): the
    re-layout kernels leave the forward chain; each conv waits on its own event (``use_events=False``: the main stream
    joins the side stream instead - inside a captured graph).
    ``epoch_ahead``: the caller will bump the parameter epoch this many times before the layouts are used (early
    segment steps run before ``FlatAdam.step`` does).  ``after``: an event (the segment's Adam launch) - the re-layouts then
    run on the prep stream behind it instead of on the side stream.  ``pingpong``: write into the conv's spare pair of buffers and swap -
    input-gradient kernels of the CURRENT step that are still to be enqueued keep reading the pair they were given."""
    if not convs:
        return
    dev = convs[0].weight.device
    main, side = _stream(), side_stream(dev)
    if _head and _PREFETCH_HEAD > 0 and len(convs) > 2 * _PREFETCH_HEAD:
        # the forward that follows waits for its FIRST layers' layouts: give those their own (small) amax launch instead of
        # queueing them behind the amax pass over all 158 MB of weights (105 us before the next step could start)
        prefetch_weight_layouts(convs[:_PREFETCH_HEAD], use_events, epoch_ahead, pingpong, after, False)
        return prefetch_weight_layouts(convs[_PREFETCH_HEAD:], use_events, epoch_ahead, pingpong, after, False)
    if after is not None and use_events:
        side = prep_stream(dev)
        after.wait(side)
    else:
        stream_wait(side, main)
    ev = None
    cache0 = convs[0].__dict__.setdefault("_wsdl_cache", {})
    with torch.cuda.stream(side):
        amaxes = multi_amax([m.weight for m in convs], persistent=True) if CONV_ARITH[0] == 1 else None
        batch = []          # convolutions whose existing split layouts are rewritten by ONE launch (after the loop)
        for i, m in enumerate(convs):
            cache = m.__dict__.setdefault("_wsdl_cache", {})
            fresh = cache.get("prep_layout") == LAYOUT_EPOCH[0]
            old = cache.get("prep") if fresh else None
            target = (cache.get("prep_spare") if fresh else None) if pingpong else old
            w = m.weight
            in_batch = (MULTI_PREP[0] and amaxes is not None and target is not None and target[0] is not None
                        and target[1] is not None and w.is_contiguous() and _both_split(w))
            ev = None
            if in_batch:
                wf, wd = target
                batch.append((w, wf, wd, amaxes[i:i + 1], cache))
            else:
                wf, wd = prep_weights(w, True, True, amaxes[i:i + 1] if amaxes is not None else None, reuse=target)
                if use_events:
                    ev = cache.get("own_event")         # one event per site, re-recorded step after step
                    if ev is None:
                        ev = cache["own_event"] = Event()
                    ev.record(side)
            cache["prep_key"] = (PARAM_EPOCH[0] + epoch_ahead, w._version, w.data_ptr())
            cache["prep"], cache["prep_event"] = (wf, wd), ev
            cache["prep_spare"] = old if pingpong else None
            cache["prep_layout"] = LAYOUT_EPOCH[0]
        if batch and torch.cuda.is_current_stream_capturing() and not prep_weights_multi([b[:4] for b in batch], lookup_only=True):
            # a capture cannot copy a new descriptor table to the device: launch them one by one (as before)
            for w, wf, wd, a, _c in batch:
                prep_weights(w, True, True, a, reuse=(wf, wd))
            batch = []
        if batch:
            prep_weights_multi([b[:4] for b in batch])
            if use_events:
                ev = cache0.get("batch_event")
                if ev is None:
                    ev = cache0["batch_event"] = Event()
                ev.record(side)
                for b in batch:
                    b[4]["prep_event"] = ev
            else:
                ev = None
    if not use_events:
        stream_wait(main, side)         # graph capture: no cross-replay events - the step ends with the layouts complete
    return ev                           # recorded behind the last re-layout (None without events)


def _stream_object(device, handle):
    """The torch stream object behind a raw handle of one of the library's streams (None: not one of them)."""
    for st in library_streams(device):
        if st.cuda_stream == handle:
            return st
    return None


def workspace(nbytes, device, stream=None):
    """Stream-ordered scratch: one growing buffer per (device, stream).  ``stream``: a raw handle (default: the current one).
    A buffer is allocated with ITS stream current, so that the caching allocator files it under that stream's pool: when a
    larger geometry replaces it, the old block can only be handed to later allocations of the same stream - behind the
    kernels still queued there that write it (a side-stream workspace allocated from the main stream's pool could be given
    to the next main-stream tensor while side-stream weight gradients were still filling it)."""
    cur = raw_stream(device)
    key = (device, cur if stream is None else stream)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        n = max(int(nbytes), 1 << 20)
        if stream is None or stream == cur:
            buf = torch.empty(n, dtype=torch.uint8, device=device)
        else:
            obj = _stream_object(device, stream)
            if obj is None:
                raise WsdlError("workspace: a stream handle that is not one of the library's streams")
            with torch.cuda.stream(obj):
                buf = torch.empty(n, dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


_size_cache = {}      # (query, geometry) -> bytes: pure functions of the geometry under a fixed option set (cleared by set_option)


def _ws_bytes(query, *geom):
    key = (query, geom)
    n = _size_cache.get(key)
    if n is None:
        n = _size_cache[key] = getattr(lib(), query)(*geom)
    return n


_coop_cache = {}
BN_COOP = [os.environ.get("WSDL_BN_COOP", "1") != "0"]


def coop_counters(device):
    """The zero-initialised counters of the several-workgroups-per-channel BatchNorm kernels (include/wsdl_hip.h
    wsdl_bn_train_fwd ``coop``): one region per (device, stream) - launches of one stream never overlap, and every launch
    leaves its counters zeroed.  None switches the form off (WSDL_BN_COOP=0)."""
    if not BN_COOP[0] or torch.cuda.is_current_stream_capturing() and (device, raw_stream(device)) not in _coop_cache:
        return None
    key = (device, raw_stream(device))
    buf = _coop_cache.get(key)
    if buf is None:
        buf = _coop_cache[key] = torch.zeros(2 * 4096, dtype=torch.int32, device=device)
    return buf


def conv_out_hw(H, W, k, stride, pad, dil):
    return ((H + 2 * pad - dil * (k - 1) - 1) // stride + 1, (W + 2 * pad - dil * (k - 1) - 1) // stride + 1)


# ------------------------------------------------------------------------------------------ raw ops
CONV_ARITH = [1]       # mirrors the library's "conv_arith": 1 = fp16x2 split kernels (need per-tensor amax scalars)


def set_option(name, value):
    """wsdl_set_option + invalidation of every cached weight layout (options such as "conv_split" change what the
    layout buffers hold)."""
    check(lib().wsdl_set_option(name.encode(), int(value)))
    if name == "conv_arith":
        CONV_ARITH[0] = int(value != 0)
    if name == "bn_coop":
        BN_COOP_ON[0] = int(value) > 0
    _both_split_cache.clear()          # what the library answered under the old option set
    _size_cache.clear()
    LAYOUT_EPOCH[0] += 1
    bump_param_epoch()


def launch_trace(on):
    """wsdl_launch_trace: the convolution entry points describe every launch they choose (form, tile, K slices, XCD order,
    bands, grid) - read back with ``last_launches()``."""
    check(lib().wsdl_launch_trace(int(bool(on))))


def last_launches():
    """This thread's launch descriptions since the previous call (a "; "-separated string)."""
    return lib().wsdl_last_launches().decode()


# ---- per-tensor amax scalars ------------------------------------------------------------------------------------
# The fp16x2 convolution kernels scale each operand tensor by a power of two derived from a device scalar
# amax >= max|tensor| (conv_split.h).  The kernel that WRITES a tensor publishes it for free (BatchNorm forward /
# backward, the eval-mode conv epilogue: an atomicMax into a zeroed slot); the tensor carries it as ``_wsdl_amax``.
# Tensors that arrive without one (the network input, concatenations, dropout outputs, views) get a read pass
# (``wsdl_amax``) the first time a split convolution consumes them.
_amax_pools = {}
AMAX_POOL_SLOTS = [4096]      # slots per pool (tests shrink it to force roll-overs)


def amax_slot(device):
    """A zero-initialised one-element fp32 view (slots are handed out once; a pool of 4096 is one memset).

    One pool per (device, STREAM), like ``workspace()``: the pool's memset is ordered on the stream that allocates it,
    and a slot's first use - the producing kernel's atomicMax into the zeroed slot - is enqueued on that same stream,
    so a roll-over on one stream (a CAM lane, the side stream's aux head) can neither wipe nor pre-date a slot another
    stream is publishing into.  Consumers on other streams read a slot only behind the stream join that orders them
    after its producer (the weight-gradient kernels on the side stream: ``side.wait_stream(main)``)."""
    key = (device, raw_stream(device))
    pool = _amax_pools.get(key)
    if pool is None or pool[1] >= AMAX_POOL_SLOTS[0]:
        # a slot is a PAIR of floats: [max|tensor|, ~bits of the smallest non-zero channel maximum] - the second one is written
        # by the BatchNorm kernels only (range sentinel, include/wsdl_hip.h wsdl_range_check); callers see the first
        buf = memset_zero(torch.empty(2 * AMAX_POOL_SLOTS[0], device=device, dtype=torch.float32))
        if not torch.cuda.is_current_stream_capturing():
            # slots are read by kernels on the other streams of this library (weight gradients on the side stream, the
            # main stream joining a CAM lane): keep the caching allocator from recycling a retired pool under them
            for st in library_streams(device):
                buf.record_stream(st)
            buf.record_stream(torch.cuda.default_stream(device))
        pool = [buf, 0]
        _amax_pools[key] = pool
    i = pool[1]
    pool[1] += 1
    return pool[0][2 * i:2 * i + 1]


RANGE_LIMIT_LOG2 = 25          # a tensor whose channel maxima spread further than 2^25 leaves the fp16x2 arithmetic's safe range
_range_out = {}
_range_rows = {}               # device -> rows of _range_out the latest range_check writes
_RANGE_ROWS = 8
_range_warned_pools = [False]


def range_check(device):
    """Reduce the (max, min channel maximum) pairs of the amax slots handed out on ``device`` since their pools were created
    (launches of the library on the current stream; no host synchronisation): ``range_status`` reads the result later."""
    device = _norm_device(device)
    pools = [(k, v) for k, v in _amax_pools.items() if k[0] == device and v[1] > 0]
    if not pools:
        return
    out = _range_out.get(device)
    if out is None:
        out = _range_out[device] = torch.zeros(_RANGE_ROWS, 3, dtype=torch.float32).pin_memory()
    if len(pools) > _RANGE_ROWS and not _range_warned_pools[0]:
        _range_warned_pools[0] = True
        import warnings
        warnings.warn(f"range_check: {len(pools)} amax pools on {device}, only the first {_RANGE_ROWS} are checked "
                      "(one pool per stream that requested slots: more streams than the library creates)")
    pools = pools[:_RANGE_ROWS]
    # the check kernels run on the CURRENT stream: a pool that belongs to another stream (the side stream's aux head, a LayerCAM
    # lane) is only read once that stream has been joined - FlatAdam.step() calls this behind join_side_stream; a pool of a
    # stream the caller did not join may still be written while it is read, so it is left out
    cur = raw_stream(device)
    joined = {cur, _side_streams[device].cuda_stream if device in _side_streams else cur}
    pools = [(k, v) for k, v in pools if k[1] in joined]
    for j, (_k, (buf, used)) in enumerate(pools):
        check(lib().wsdl_range_check(_p(buf), int(used), RANGE_LIMIT_LOG2, _p(out[j]), _stream()))
    _range_rows[device] = len(pools)      # rows beyond hold an earlier call's figures: range_status does not read them


def range_status(device):
    """What the last completed ``range_check`` found: {"worst_log2", "pairs_over_limit", "pairs_seen", "exceeded"} - read from
    host memory the check kernels write; a step old at most."""
    out = _range_out.get(_norm_device(device))
    if out is None:
        return {"worst_log2": 0.0, "pairs_over_limit": 0, "pairs_seen": 0, "exceeded": False, "limit_log2": RANGE_LIMIT_LOG2}
    a = out.numpy()[:_range_rows.get(_norm_device(device), 0)]      # a view of the pinned buffer (polled after every replayed step: a few microseconds)
    if a.shape[0] == 0:
        return {"worst_log2": 0.0, "pairs_over_limit": 0, "pairs_seen": 0, "exceeded": False, "limit_log2": RANGE_LIMIT_LOG2}
    over = int(a[:, 1].sum())
    return {"worst_log2": float(a[:, 0].max()), "pairs_over_limit": over, "pairs_seen": int(a[:, 2].sum()), "exceeded": over > 0,
            "limit_log2": RANGE_LIMIT_LOG2}


_lane_streams = {}     # device -> the library's streams beyond the side and the prep stream (LayerCAM lanes 2, 3, ...)


def lane_stream(device, i):
    """The i-th extra stream of the library on ``device``.  A ROCm process has four hardware queues by default and HIP
    streams beyond them share one (their work then runs one behind the other): a training step that has run in the process
    holds the side stream (and possibly the prep stream), so LayerCAM lanes with streams of their own made three batches in
    flight take 0.27 ms/img instead of 0.185.  Lanes 0 and 1 therefore ARE the side and the prep stream; only further
    lanes create streams, once per device whatever the number of generators."""
    device = _norm_device(device)
    if i == 0:
        return side_stream(device)
    if i == 1:
        return prep_stream(device)
    more = _lane_streams.setdefault(device, [])
    while len(more) < i - 1:
        if stream_census(device)["total"] >= MAX_HW_QUEUES:
            # no fifth stream: it would share a hardware queue with one of the four (and, in a data-parallel process, slow the
            # gradient collectives or the weight gradients down with it).  Further lanes take turns on the side / prep stream.
            return (side_stream(device), prep_stream(device))[i % 2]
        more.append(torch.cuda.Stream(device=device))
    return more[i - 2]


MAX_HW_QUEUES = int(os.environ.get("GPU_MAX_HW_QUEUES", "4"))     # hardware queues of a ROCm process (the runtime's own switch)


def stream_census(device):
    """The streams this process keeps busy on ``device``: torch's current (default) stream, the library's side / prep / lane
    streams, and - once a process group with the nccl (RCCL) backend exists - the stream RCCL enqueues its collectives on.
    A ROCm process has ``MAX_HW_QUEUES`` hardware queues (4); a stream beyond them shares one, i.e. runs behind another
    stream's work (measured: three LayerCAM lanes 0.185 -> 0.27 ms/img with a fifth stream in use, profiles/r03_notes.md).
    ``lane_stream`` refuses to create the fifth; tests/test_hip_dp.py checks the count in a data-parallel process."""
    device = _norm_device(device)
    c = {"main": 1, "side": int(device in _side_streams), "prep": int(device in _prep_streams),
         "lanes": len(_lane_streams.get(device, [])), "rccl": 0}
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_backend() == "nccl":
            c["rccl"] = 1
    except Exception:        # a process group that is being torn down
        pass
    c["total"] = sum(c.values())
    c["hw_queues"] = MAX_HW_QUEUES
    return c


def library_streams(device):
    """Every stream this library has created on ``device`` (amax slots may be read on any of them)."""
    device = _norm_device(device)
    return [st for st in (_side_streams.get(device), _prep_streams.get(device)) if st is not None] + list(_lane_streams.get(device, []))


def _publish_amax(t, slot):
    """Attach a producer-published amax scalar to the tensor it bounds, with the tensor's version: a later in-place
    modification (``t.mul_()``, ``buf.copy_()``) makes ``amax_of`` measure again instead of trusting a stale bound."""
    t._wsdl_amax = slot
    t._wsdl_amax_version = t._version


def reset_amax_pool(device):
    """Forget the current slot pools of ``device``: the next request allocates (and zeroes) a new one - graph.py brackets a
    capture with it so that the captured step's slots are re-zeroed by every replay."""
    device = device if isinstance(device, torch.device) else torch.device(device)
    for key in [k for k in _amax_pools if k[0] == device]:
        _amax_pools.pop(key, None)


def _split_kc(kc, taps):
    """Could a convolution contracting ``kc`` channels over ``taps`` taps run on the split kernels? (split_eligible in
    conv_igemm.hip; an over-approximation only costs an unused amax)."""
    return CONV_ARITH[0] == 1 and kc % 16 == 0 and taps <= 9


def amax_of(t, needed=True):
    """The tensor's amax scalar: the one its producer published, else one read pass (cached on the tensor)."""
    if not needed:
        return None
    a = getattr(t, "_wsdl_amax", None)
    if a is not None and getattr(t, "_wsdl_amax_version", t._version) != t._version:
        a = None                 # modified in place since the bound was taken (copy_ into a reused buffer, mul_): measure again
    if a is None:
        tt, bs = _planes(t, "amax input") if t.dim() == 4 else (_dense(t, "amax input"), 0)
        a = amax_slot(t.device)
        B = tt.shape[0] if tt.dim() == 4 else 1
        per = tt.numel() // B
        check(lib().wsdl_amax(_p(tt), B, per, bs if tt.dim() == 4 else per, _p(a), 0, _stream()))    # the slot is zeroed
        try:
            t._wsdl_amax = a
            t._wsdl_amax_version = t._version
        except AttributeError:
            pass
    return a


def _layout_buffer(w, dgrad):
    import ctypes
    Cout, Cin, kh, kw = w.shape
    plain = ctypes.c_int(0)
    nbytes = lib().wsdl_conv2d_weight_layout_bytes(Cout, Cin, kh, kw, int(dgrad), ctypes.byref(plain))
    if plain.value and dgrad and kh * kw == 1:
        return w.detach().reshape(Cout, Cin), False        # [tap*Cout+co][ci] of a 1x1 kernel IS w's own layout
    return torch.empty(nbytes // 4, device=w.device, dtype=torch.float32), True


def prep_weights(w, want_fwd=True, want_dgrad=True, w_amax=None, reuse=None):
    """Opaque layout buffers (wt_fwd, wt_dgrad) for conv2d_fwd / conv2d_dgrad (include/wsdl_hip.h).  ``w_amax``: device
    scalar max|w| if the caller already has it (``multi_amax``); otherwise the library reduces it.  ``reuse``: a
    previous (wt_fwd, wt_dgrad) pair of this weight to overwrite in place (the per-step re-layout keeps its buffers:
    no allocator churn, and a captured graph keeps reading the addresses it was captured with)."""
    w = _dense(w, "weight")
    Cout, Cin, kh, kw = w.shape
    if reuse is not None and reuse[0] is not None and (reuse[1] is not None or not want_dgrad):
        wf, wd = reuse
        make_wd = wd is not None and wd.data_ptr() != w.data_ptr()
        check(lib().wsdl_conv2d_prep_weights(_p(w), _p(wf), _p(wd if make_wd else None), Cout, Cin, kh, kw, _p(w_amax),
                                             _stream()))
        return wf, wd
    wf = _layout_buffer(w, False)[0] if want_fwd else None
    wd, make_wd = _layout_buffer(w, True) if want_dgrad else (None, False)
    if want_fwd or make_wd:
        check(lib().wsdl_conv2d_prep_weights(_p(w), _p(wf), _p(wd if make_wd else None), Cout, Cin, kh, kw, _p(w_amax),
                                             _stream()))
    return wf, wd


_multi_amax_cache = {}


def multi_amax(tensors, persistent=False):
    """max|t| of every tensor of a FIXED list in one launch -> (n,) device tensor.  The pointer / count tables live on
    the device and are built once per list (parameters keep their addresses inside the flat optimiser buffer).
    ``persistent``: the result goes into one buffer per list, overwritten by the next call (stream-ordered consumers only)."""
    key = tuple((t.data_ptr(), t.numel()) for t in tensors)
    ent = _multi_amax_cache.get(key)
    dev = tensors[0].device
    if ent is None:
        ptrs = torch.tensor([k[0] for k in key], dtype=torch.int64).to(dev)
        counts = torch.tensor([k[1] for k in key], dtype=torch.int64).to(dev)
        ent = _multi_amax_cache[key] = (ptrs, counts, torch.empty(len(tensors), device=dev, dtype=torch.float32))
        if len(_multi_amax_cache) > 64:
            _multi_amax_cache.pop(next(iter(_multi_amax_cache)))
    out = ent[2] if persistent else torch.empty(len(tensors), device=dev, dtype=torch.float32)
    check(lib().wsdl_multi_amax(_p(ent[0]), _p(ent[1]), len(tensors), _p(out), _stream()))
    return out


MULTI_PREP = [os.environ.get("WSDL_MULTI_PREP", "1") != "0"]      # the per-step re-layout of all split layouts in one launch
_both_split_cache = {}
_prep_tables = {}


def _both_split(w):
    """Are the forward and the dgrad layout of this weight both split layouts (what wsdl_conv2d_prep_weights_multi takes)?"""
    import ctypes
    key = tuple(w.shape)
    r = _both_split_cache.get(key)
    if r is None:
        Cout, Cin, kh, kw = key
        pf, pd = ctypes.c_int(0), ctypes.c_int(0)
        lib().wsdl_conv2d_weight_layout_bytes(Cout, Cin, kh, kw, 0, ctypes.byref(pf))
        lib().wsdl_conv2d_weight_layout_bytes(Cout, Cin, kh, kw, 1, ctypes.byref(pd))
        r = _both_split_cache[key] = (pf.value == 0 and pd.value == 0 and kh * kw <= 9)
    return r


class _PrepDesc(C.Structure):
    _fields_ = [("w", C.c_void_p), ("wt_fwd", C.c_void_p), ("wt_dgrad", C.c_void_p), ("w_amax", C.c_void_p),
                ("Cout", C.c_int), ("Cin", C.c_int), ("taps", C.c_int), ("grid_x", C.c_int),
                ("block_begin", C.c_int), ("reserved", C.c_int)]


def prep_weights_multi(entries, lookup_only=False):
    """entries: [(weight, wt_fwd, wt_dgrad, w_amax (1,) device tensor)] with existing layout buffers, all ``_both_split``:
    one launch re-lays them all out (include/wsdl_hip.h wsdl_conv2d_prep_weights_multi).  The descriptor table goes to the
    device once per set of addresses (weights live in the flat optimiser buffer, layout buffers are reused step after step)."""
    key = tuple((w.data_ptr(), wf.data_ptr(), wd.data_ptr(), a.data_ptr()) for w, wf, wd, a in entries)
    ent = _prep_tables.get(key)
    if lookup_only:
        return ent is not None
    if ent is None:
        arr = (_PrepDesc * len(entries))()
        blocks = 0
        for d, (w, wf, wd, a) in zip(arr, entries):
            Cout, Cin, kh, kw = w.shape
            d.w, d.wt_fwd, d.wt_dgrad, d.w_amax = w.data_ptr(), wf.data_ptr(), wd.data_ptr(), a.data_ptr()
            d.Cout, d.Cin, d.taps, d.grid_x, d.block_begin = Cout, Cin, kh * kw, (Cin + 31) // 32, blocks
            blocks += d.grid_x * ((Cout + 31) // 32)
        table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(entries[0][0].device)
        ent = _prep_tables[key] = (table, len(entries), blocks)
        if len(_prep_tables) > 64:
            _prep_tables.pop(next(iter(_prep_tables)))
    check(lib().wsdl_conv2d_prep_weights_multi(_p(ent[0]), ent[1], ent[2], _stream()))


def conv2d_fwd(x, wt_fwd, wshape, stride, pad, dil, scale=None, shift=None, residual=None, relu=False, out=None,
               x_amax=None, want_amax=False):
    """``x_amax``: the input's amax scalar (looked up / computed when None and the split kernels may run);
    ``want_amax``: publish max|out| as ``out._wsdl_amax`` (the output feeds another convolution directly)."""
    if x_amax is None:
        x_amax = amax_of(x, _split_kc(x.shape[1], wshape[2] * wshape[3]))
    x, x_bs = _planes(x, "x")
    B, Cin, H, W = x.shape
    Cout, Cin2, kh, kw = wshape
    if Cin2 != Cin:
        raise WsdlError(f"conv2d: input has {Cin} channels, weight expects {Cin2}")
    OH, OW = conv_out_hw(H, W, kh, stride, pad, dil)
    if OH <= 0 or OW <= 0:
        raise WsdlError(f"conv2d: empty output ({OH} x {OW})")
    if out is None:
        out = torch.empty(B, Cout, OH, OW, device=x.device, dtype=torch.float32)
        y_bs = Cout * OH * OW
    else:
        out, y_bs = _planes(out, "out")
        if tuple(out.shape) != (B, Cout, OH, OW):
            raise WsdlError("conv2d: bad preallocated output shape")
    res_bs = 0
    if residual is not None:
        residual, res_bs = _planes(residual, "residual")
        if tuple(residual.shape) != (B, Cout, OH, OW):
            raise WsdlError("conv2d: residual shape mismatch")
    nws = _ws_bytes("wsdl_conv2d_igemm_workspace", B, Cin, H, W, Cout, kh, kw, stride, pad, dil, 0)
    ws = workspace(nws, x.device) if nws else None
    y_amax = amax_slot(x.device) if (want_amax and CONV_ARITH[0] == 1) else None
    check(lib().wsdl_conv2d_fwd(_p(x), _p(wt_fwd), _p(out), B, Cin, H, W, Cout, kh, kw, stride, pad, dil,
                                _p(scale), _p(shift), _p(residual), int(relu), x_bs, y_bs, res_bs, _p(x_amax), _p(y_amax),
                                _p(ws), ws.numel() if ws is not None else 0, _stream()))
    if y_amax is not None:
        _publish_amax(out, y_amax)
    return out


def conv2d_dgrad(dy, wt_dgrad, wshape, xshape, stride, pad, dil, accumulate_into=None, dy_amax=None, acc_mask=None):
    """``accumulate_into``: dx = dgrad + that tensor, in place.  ``acc_mask`` (bits from ``bn_train_fwd(want_mask=True)``, same
    shape as dx): only the elements of ``accumulate_into`` whose bit is set count (the masked gradient of an identity branch)."""
    if acc_mask is not None and (accumulate_into is None or acc_mask.dtype != torch.uint8
                                 or acc_mask.numel() * 8 != accumulate_into.numel() or not accumulate_into.is_contiguous()):
        raise WsdlError("conv2d_dgrad: acc_mask needs a dense accumulate_into tensor of 8 x its number of bytes")
    if dy_amax is None:
        dy_amax = amax_of(dy, _split_kc(wshape[0], wshape[2] * wshape[3]))
    dy, dy_bs = _planes(dy, "dy")
    B, Cin, H, W = xshape
    Cout, _, kh, kw = wshape
    dx = accumulate_into if accumulate_into is not None else torch.empty(xshape, device=dy.device, dtype=torch.float32)
    nws = _ws_bytes("wsdl_conv2d_igemm_workspace", B, Cin, H, W, Cout, kh, kw, stride, pad, dil, 1)
    ws = workspace(nws, dy.device) if nws else None
    check(lib().wsdl_conv2d_dgrad(_p(dy), _p(wt_dgrad), _p(dx), B, Cin, H, W, Cout, kh, kw, stride, pad, dil,
                                  int(accumulate_into is not None), _p(acc_mask), dy_bs, _p(dy_amax), _p(ws),
                                  ws.numel() if ws is not None else 0, _stream()))
    dx._wsdl_fresh = True        # a buffer this library has just produced and nobody else holds (see _owned)
    return dx


def fwd_group_ok(n, xshape, cout):
    """Does the library serve ``n`` forward convolutions of one input of ``xshape`` as one grouped launch?"""
    B, Cin, H, W = xshape
    return bool(lib().wsdl_conv2d_fwd_group_ok(int(n), B, Cin, H, W, int(cout)))


def conv2d_fwd_group(x, wfs, wshapes, dils, x_amax=None, outs=None):
    """[conv(x, w_i)] for 1x1 / 3x3 stride-1 'same' convolutions of ONE input in one launch (include/wsdl_hip.h
    wsdl_conv2d_fwd_group): raw outputs, no epilogue."""
    n = len(wfs)
    if x_amax is None:
        x_amax = amax_of(x, _split_kc(x.shape[1], 9))
    x, x_bs = _planes(x, "x")
    B, Cin, H, W = x.shape
    Cout = wshapes[0][0]
    if any(ws[0] != Cout or ws[1] != Cin for ws in wshapes):
        raise WsdlError("conv2d_fwd_group: the problems must share their channel counts")
    IA, PA, LA = C.c_int * n, _vp * n, C.c_longlong * n
    ks, ds = IA(*[int(ws[2]) for ws in wshapes]), IA(*[int(d) for d in dils])
    nws = lib().wsdl_conv2d_fwd_group_workspace(n, ks, ds, B, Cin, H, W, Cout)
    if nws == 0:
        raise WsdlError("conv2d_fwd_group: geometry not supported")
    ws = workspace(nws, x.device)
    if outs is None:
        outs = [torch.empty(B, Cout, H, W, device=x.device, dtype=torch.float32) for _ in range(n)]
    bss = []
    dense = []
    for o in outs:
        t, bs = _planes(o, "out")
        if t is not o:
            raise WsdlError("conv2d_fwd_group: preallocated outputs must be dense planes")
        dense.append(t)
        bss.append(bs)
    check(lib().wsdl_conv2d_fwd_group(n, _p(x), PA(*[_p(w) for w in wfs]), PA(*[_p(t) for t in dense]), ks, ds, B, Cin, H, W,
                                      Cout, x_bs, LA(*bss), _p(x_amax), _p(ws), ws.numel(), _stream()))
    return outs


def dgrad_multi_ok(n, xshape, cout):
    """Does the library serve the one-launch input gradient of ``n`` convolutions over an input of ``xshape``?"""
    B, Cin, H, W = xshape
    return bool(lib().wsdl_conv2d_dgrad_multi_ok(int(n), B, Cin, H, W, int(cout)))


def conv2d_dgrad_multi(dys, wds, wshapes, dils, xshape, accumulate_into=None, dy_amaxes=None):
    """dx = [accumulate_into +] sum_i dgrad(dys[i], wds[i]) in ONE launch (include/wsdl_hip.h wsdl_conv2d_dgrad_multi):
    convolutions of the same input, 1x1 / 3x3, stride 1, 'same' padding.  The first source decides the column bands of
    the padding-tap skipping."""
    n = len(dys)
    B, Cin, H, W = xshape
    Cout = wshapes[0][0]
    amaxes = []
    dense = []
    bss = []
    for i in range(n):
        if wshapes[i][0] != Cout or wshapes[i][1] != Cin:
            raise WsdlError("conv2d_dgrad_multi: the sources must share their channel counts")
        a = dy_amaxes[i] if dy_amaxes is not None and dy_amaxes[i] is not None else \
            amax_of(dys[i], _split_kc(Cout, wshapes[i][2] * wshapes[i][3]))
        t, bs = _planes(dys[i], "dy")
        amaxes.append(a)
        dense.append(t)
        bss.append(bs)
    dx = accumulate_into if accumulate_into is not None else torch.empty(xshape, device=dys[0].device, dtype=torch.float32)
    PA, IA, LA = _vp * n, C.c_int * n, C.c_longlong * n
    check(lib().wsdl_conv2d_dgrad_multi(n, PA(*[_p(t) for t in dense]), PA(*[_p(w) for w in wds]), PA(*[_p(a) for a in amaxes]),
                                        IA(*[int(ws[2]) for ws in wshapes]), IA(*[int(d) for d in dils]), LA(*bss), _p(dx),
                                        B, Cin, H, W, Cout, int(accumulate_into is not None), _stream()))
    dx._wsdl_fresh = True
    return dx


def _wgrad_split(wshape):
    """May the library's fp16x2 weight-gradient kernel take this shape (then both operands need amax scalars - a read
    pass on the main stream for a tensor that carries none, so this must not claim more than the library's own rule)."""
    return CONV_ARITH[0] == 1 and wshape[0] % 128 == 0 and wshape[1] % 128 == 0


class _ReduceDesc(C.Structure):       # wsdl_wgrad_reduce_desc (include/wsdl_hip.h)
    _fields_ = [("slab", C.c_void_p), ("dw", C.c_void_p), ("live", C.c_ulonglong),
                ("S", C.c_int), ("Cout", C.c_int), ("Cin", C.c_int), ("T", C.c_int),
                ("accumulate", C.c_int), ("kind", C.c_int), ("grid_x", C.c_int), ("nblocks", C.c_int),
                ("block_begin", C.c_int), ("reserved", C.c_int)]


# Deferred slab reductions of the weight gradients (wsdl_conv2d_wgrad_deferred / wsdl_wgrad_reduce_multi): a weight gradient
# written into a parameter's slice of the flat gradient buffer leaves its pixel slabs un-reduced; the pending reductions run
# as one launch when the gradients are needed - at the end of the backward pass (an autograd engine callback), before a
# gradient bucket's all-reduce, before the optimiser step - or in groups (WSDL_WGRAD_DEFER_MB: flushed once the pending slabs
# exceed that many MB).  Bit-identical to the per-layer reductions.
# BUILT, MEASURED, OFF BY DEFAULT (round 6; WSDL_WGRAD_DEFER=1 switches it on).  ~60 launches of 9-10 us per training step do
# become 1-10, but the step is SLOWER: 853.8 / 852.5 img/s in 48 MB groups, 846 in 16 MB groups, 854.6 / 856.4 all at the end,
# against 864.0 / 864.7 with the per-layer launches (same box, profiles/r06_notes.md).  A layer's slabs are reduced by the
# launch right behind the kernel that wrote them - out of the L2s and the Infinity Cache, from ONE workspace the next layer
# reuses; deferred, every layer needs slabs of its own (0.5 GB per step) that go out to HBM and come back.  The launches
# saved were latency on the side stream, which is not what bounds the step.
WGRAD_DEFER = [os.environ.get("WSDL_WGRAD_DEFER", "0") != "0"]
WGRAD_DEFER_BYTES = [int(float(os.environ.get("WSDL_WGRAD_DEFER_MB", "100000")) * (1 << 20))]
_wgrad_pending = {}       # device -> {"descs": [_ReduceDesc], "dws": set of dw pointers, "keep": [tensors], "side": bool, "cb": bool}
_wgrad_ws = {}            # (device, geometry, dw pointer) -> that layer's own workspace (its slabs outlive the launch)
_reduce_tables = {}       # descriptor bytes -> (device table, n, total blocks)


def _pending_of(device):
    pend = _wgrad_pending.get(device)
    if pend is None:
        pend = _wgrad_pending[device] = {"descs": [], "dws": set(), "keep": [], "side": False, "cb": False, "bytes": 0}
    return pend


def flush_wgrad_reduces(device=None):
    """Run every pending slab reduction of ``device`` (default: all devices) as ONE launch - on the side stream, behind what the
    current stream holds, when any of the weight gradients ran there (consumers of the gradients join the side stream anyway:
    ``join_side_stream`` before Adam, the bucket all-reduces are enqueued on it)."""
    for dev, pend in list(_wgrad_pending.items()):
        if (device is not None and _norm_device(device) != dev) or not pend["descs"]:
            continue
        descs = pend["descs"]
        key = b"".join(bytes(d) for d in descs)
        ent = _reduce_tables.get(key)
        if ent is None:
            arr = (_ReduceDesc * len(descs))()
            blocks = 0
            for a, d in zip(arr, descs):
                C.memmove(C.addressof(a), C.addressof(d), C.sizeof(_ReduceDesc))
                a.block_begin = blocks
                blocks += d.nblocks
            with torch.cuda.device(dev):
                table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
            ent = _reduce_tables[key] = (table, len(descs), blocks)
            if len(_reduce_tables) > 64:
                _reduce_tables.pop(next(iter(_reduce_tables)))
        cur = raw_stream(dev)
        if pend["side"]:
            hs = side_stream(dev).cuda_stream
            stream_wait(hs, cur)
        else:
            hs = cur
        rec = PLAN_REC[0]
        if rec is not None:
            rec.keep.extend(pend["keep"])        # the slabs' workspaces: their addresses are inside the table, not arguments
        check(lib().wsdl_wgrad_reduce_multi(_p(ent[0]), ent[1], ent[2], hs))
        pend["descs"], pend["keep"], pend["side"], pend["cb"], pend["bytes"] = [], [], False, False, 0
        pend["dws"].clear()


def _queue_flush(pend):
    """Flush when the running backward pass ends (the engine's final callbacks - what DistributedDataParallel uses for its
    own finalisation); outside a backward pass, at once."""
    if pend["cb"]:
        return True
    try:
        torch.autograd.Variable._execution_engine.queue_callback(flush_wgrad_reduces)
    except RuntimeError:
        return False
    pend["cb"] = True
    return True


def wgrad_presplit_bytes(xshape, wshape, stride, pad, dil):
    """Bytes of the pre-split dY rows the weight gradient of this convolution reads when its producer writes them
    (``bn_train_bwd(presplit_bytes=)``); 0 where it would not use them."""
    B, Cin, H, W = xshape
    Cout, _, kh, kw = wshape
    return _ws_bytes("wsdl_conv2d_wgrad_presplit_bytes", B, Cin, H, W, Cout, kh, kw, stride, pad, dil)


def conv2d_wgrad(x, dy, wshape, stride, pad, dil, out=None, accumulate=False, x_amax=None, dy_amax=None, stream=None, defer=False,
                 x_camax=None, dy_camax=None, dy_presplit=None):
    """``stream``: raw handle of the stream to launch on (default: the current stream) - the side-stream launches of the
    training step pass it instead of switching torch's current stream (a ``with torch.cuda.stream()`` costs the host ~20 us,
    61 times per step).  ``defer``: leave the slab reduction to ``flush_wgrad_reduces`` (``out`` given; the caller has made
    sure a flush follows - ``_wgrad_into``).  ``x_camax`` / ``dy_camax``: per-channel maxima of the operands (``_wsdl_camax`` of a
    BatchNorm output / input gradient), ``dy_presplit``: dY as the kernel's fp16 rows (``_wsdl_presplit``) - wsdl_conv2d_wgrad_ex."""
    if dy_camax is None:
        dy_presplit = None
    elif dy_presplit is None:
        dy_camax_p = getattr(dy, "_wsdl_camax", None)
        if dy_camax_p is dy_camax:
            dy_presplit = getattr(dy, "_wsdl_presplit", None)
    if stream is not None and stream != _stream():
        # everything this function would otherwise enqueue on the CURRENT stream (amax passes, densifying copies) must have
        # been resolved by the caller, in front of the wait that orders ``stream`` behind the current one (_wgrad_into)
        split = _wgrad_split(wshape)
        if out is None or (split and (x_amax is None or dy_amax is None)):
            raise WsdlError("conv2d_wgrad(stream=): pass out, x_amax and dy_amax (resolved on the current stream)")
        if _planes(x, "x")[0] is not x or _planes(dy, "dy")[0] is not dy:
            raise WsdlError("conv2d_wgrad(stream=): operands must already be dense planes")
    if x_amax is None:
        x_amax = amax_of(x, _wgrad_split(wshape))
    if dy_amax is None:
        dy_amax = amax_of(dy, _wgrad_split(wshape))
    x, x_bs = _planes(x, "x")
    dy, dy_bs = _planes(dy, "dy")
    B, Cin, H, W = x.shape
    Cout, _, kh, kw = wshape
    geom = (B, Cin, H, W, Cout, kh, kw, stride, pad, dil)
    nbytes = _ws_bytes("wsdl_conv2d_wgrad_workspace", *geom)
    if nbytes == 0:
        raise WsdlError("conv2d_wgrad: bad geometry " + repr(geom))
    if defer and out is not None:
        dev = _norm_device(x.device)
        pend = _pending_of(dev)
        if out.data_ptr() in pend["dws"]:
            flush_wgrad_reduces(dev)            # a second gradient into the same parameter: the first one's reduction goes first
        # this layer's OWN workspace: its slabs stay until the multi-reduce has run (and their addresses repeat step after step,
        # so the descriptor table is uploaded once)
        wkey = (dev, geom, out.data_ptr())
        ws = _wgrad_ws.get(wkey)
        if ws is None or ws.numel() < nbytes:
            ws = _wgrad_ws[wkey] = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=x.device)
        desc = _ReduceDesc()
        hs = _stream() if stream is None else stream
        check(lib().wsdl_conv2d_wgrad_ex(_p(x), _p(dy), _p(out), *geom, int(accumulate), x_bs, dy_bs, _p(x_amax), _p(dy_amax),
                                         _p(x_camax), _p(dy_camax), _p(dy_presplit), _p(ws), ws.numel(), C.addressof(desc), hs))
        if desc.kind >= 0:
            pend["descs"].append(desc)
            pend["dws"].add(out.data_ptr())
            pend["keep"].append(ws)
            pend["side"] = pend["side"] or hs != raw_stream(dev)
            pend["bytes"] += 4 * desc.S * desc.Cout * desc.Cin * desc.T
            if pend["bytes"] >= WGRAD_DEFER_BYTES[0] or not _queue_flush(pend):
                flush_wgrad_reduces(dev)
        return out
    ws = workspace(nbytes, x.device, stream)
    if out is None:
        out = torch.empty(wshape, device=x.device, dtype=torch.float32)
        accumulate = False
    check(lib().wsdl_conv2d_wgrad_ex(_p(x), _p(dy), _p(out), *geom, int(accumulate), x_bs, dy_bs, _p(x_amax), _p(dy_amax),
                                     _p(x_camax), _p(dy_camax), _p(dy_presplit), _p(ws), ws.numel(), None,
                                     _stream() if stream is None else stream))
    return out


def bias_grad(dy, out=None, accumulate=False):
    dy, dy_bs = _planes(dy, "dy")
    B, Cc, H, W = dy.shape
    if out is None:
        out = torch.empty(Cc, device=dy.device, dtype=torch.float32)
        accumulate = False
    check(lib().wsdl_bias_grad(_p(dy), _p(out), B, Cc, H * W, dy_bs, int(accumulate), _stream()))
    return out


def bn_fold(bn_weight, bn_bias, running_mean, running_var, eps):
    Cc = bn_weight.numel()
    scale = torch.empty(Cc, device=bn_weight.device, dtype=torch.float32)
    shift = torch.empty_like(scale)
    check(lib().wsdl_bn_fold(_p(_dense(bn_weight)), _p(_dense(bn_bias)), _p(_dense(running_mean)),
                             _p(_dense(running_var)), float(eps), _p(scale), _p(shift), Cc, _stream()))
    return scale, shift


# Per-channel maxima from the channel-resident BatchNorm kernels -> per-channel scales of the weight gradients (exact; the range
# guard of the weight gradient for nothing), and the weight gradient's dY operand written pre-split by the BatchNorm backward that
# produces it (no dy_split16 pass).  WSDL_CHAN_AMAX=0 / WSDL_DY_PRESPLIT=0: the round-5 forms (A/B).
CHAN_AMAX = [os.environ.get("WSDL_CHAN_AMAX", "1") != "0"]
DY_PRESPLIT = [os.environ.get("WSDL_DY_PRESPLIT", "1") != "0"]


def _bn_resident(B, Cc, HW, backward):
    key = ("wsdl_bn_channel_resident", (B, Cc, HW, backward))
    r = _size_cache.get(key)
    if r is None:
        r = _size_cache[key] = bool(lib().wsdl_bn_channel_resident(B, Cc, HW, backward)) and coop_off()
    return r


def coop_off():
    return not BN_COOP_ON[0]


BN_COOP_ON = [False]       # set by set_option("bn_coop", n > 0): several workgroups per channel publish no per-channel maximum


# The ReLU mask of a residual layer as bits written by the forward kernel (1/32 of the bytes of y, which the backward would
# otherwise read for it): WSDL_BN_RELU_BITS=0 reads y instead (A/B; same result bit for bit).
BN_RELU_BITS = [os.environ.get("WSDL_BN_RELU_BITS", "1") != "0"]


def bn_train_fwd(x, gamma, beta, running_mean, running_var, momentum, eps, residual=None, relu=False, out=None,
                 want_mask=False, mask_if=True, amax_into=None):
    """``want_mask``: returns a fourth value - the ReLU mask as bits (uint8, numel/8 bytes; ``bn_train_bwd(relu_mask=)``),
    or None where the kernels do not write one (no ReLU, H*W not a multiple of 8, ``mask_if`` false)."""
    x = _dense(x, "x")
    B, Cc, H, W = x.shape
    if out is None:
        out = torch.empty_like(x)
        y_bs = Cc * H * W
    else:
        out, y_bs = _planes(out, "out")
    stats = torch.empty(2, Cc, device=x.device, dtype=torch.float32)      # one allocation for the two saved statistics
    mean, invstd = stats.unbind(0)
    ws = workspace(_ws_bytes("wsdl_bn_workspace", Cc), x.device)
    if residual is not None:
        residual = _dense(residual, "residual")
    # amax_into: a slot shared by several producers (the branches of a concatenation: the maximum over all of them)
    y_amax = (amax_into if amax_into is not None else amax_slot(x.device)) if CONV_ARITH[0] == 1 else None
    mask = None
    if want_mask and mask_if and relu and BN_RELU_BITS[0] and (H * W) % 8 == 0 and y_bs % 4 == 0:
        mask = torch.empty(x.numel() // 8, device=x.device, dtype=torch.uint8)
    # per-channel maxima of the output: the scales of the weight gradient that reads it as its x operand (free in the
    # channel-resident kernel: the channel's workgroup has the maximum anyway)
    camax = None
    if y_amax is not None and CHAN_AMAX[0] and y_bs % 4 == 0 and _bn_resident(B, Cc, H * W, 0):
        camax = torch.empty(Cc, device=x.device, dtype=torch.float32)
    check(lib().wsdl_bn_train_fwd(_p(x), _p(gamma), _p(beta), _p(out), _p(mean), _p(invstd), _p(running_mean),
                                  _p(running_var), float(momentum), float(eps), B, Cc, H * W, _p(residual),
                                  int(relu), y_bs, _p(y_amax), _p(mask), _p(ws), ws.numel(), _p(coop_counters(x.device)),
                                  _p(camax), _stream()))
    if y_amax is not None:
        _publish_amax(out, y_amax)
    if camax is not None:
        out._wsdl_camax = camax
    return (out, mean, invstd, mask) if want_mask else (out, mean, invstd)


def bn_train_bwd(x, dy, y, gamma, mean, invstd, relu, want_dres, dgamma_out=None, dbeta_out=None, accumulate=False,
                 beta=None, relu_mask=None, presplit_bytes=0):
    """``relu``: True with ``y`` = the forward output (mask read from it), or True with ``relu_mask`` = the bits the
    forward wrote (``bn_train_fwd(want_mask=True)``), or True with ``y=None`` and ``beta`` given: the mask is recomputed
    from x (no residual was added in the forward) and y is never touched."""
    x = _dense(x, "x")
    dy, dy_bs = _planes(dy, "dy")
    B, Cc, H, W = x.shape
    y_bs = 0
    mode = 0
    if relu:
        if relu_mask is not None:
            if relu_mask.dtype != torch.uint8 or relu_mask.numel() * 8 != x.numel() or not relu_mask.is_cuda:
                raise WsdlError("bn_train_bwd: relu_mask must be the uint8 bit mask of the forward (numel / 8 bytes)")
            mode = 3
        elif y is None:
            if beta is None:
                raise WsdlError("bn_train_bwd: relu needs the forward output, its bit mask or beta")
            mode = 2
        else:
            mode = 1
            y, y_bs = _planes(y, "y")
    dx = torch.empty_like(x)
    acc = bool(accumulate) and dgamma_out is not None and dbeta_out is not None
    dgamma = dgamma_out if dgamma_out is not None else torch.empty(Cc, device=x.device, dtype=torch.float32)
    dbeta = dbeta_out if dbeta_out is not None else torch.empty(Cc, device=x.device, dtype=torch.float32)
    dres = torch.empty_like(x) if want_dres else None
    ws = workspace(_ws_bytes("wsdl_bn_workspace", Cc), x.device)
    dx_amax = amax_slot(x.device) if CONV_ARITH[0] == 1 else None
    # ``presplit_bytes`` (wsdl_conv2d_wgrad_presplit_bytes of the convolution whose output gradient this is): dx is also written
    # as the fp16 rows that convolution's weight gradient reads, scaled per channel - dy_split16_kernel does not run for it
    camax = presplit = None
    if dx_amax is not None and CHAN_AMAX[0] and dy_bs % 4 == 0 and y_bs % 4 == 0 and _bn_resident(B, Cc, H * W, 1):
        camax = torch.empty(Cc, device=x.device, dtype=torch.float32)
        if presplit_bytes and DY_PRESPLIT[0] and (B * H * W) % 32 == 0:
            presplit = torch.empty(int(presplit_bytes), device=x.device, dtype=torch.uint8)
    check(lib().wsdl_bn_train_bwd(_p(x), _p(dy), _p(y if mode == 1 else None), _p(gamma),
                                  _p(_dense(beta) if mode == 2 else None), _p(mean), _p(invstd), _p(dx),
                                  _p(dgamma), _p(dbeta), _p(dres), B, Cc, H * W, mode, int(acc), dy_bs, y_bs,
                                  _p(dx_amax), _p(relu_mask if mode == 3 else None), _p(ws), ws.numel(),
                                  _p(coop_counters(x.device)), _p(camax), _p(presplit), _stream()))
    if dx_amax is not None:
        _publish_amax(dx, dx_amax)
    if camax is not None:
        dx._wsdl_camax = camax
        if presplit is not None:
            dx._wsdl_presplit = presplit
    dx._wsdl_fresh = True
    if dres is not None:
        dres._wsdl_fresh = True
    return dx, dgamma, dbeta, dres


def _owned(t):
    """May this incoming gradient buffer be accumulated into IN PLACE?  Only buffers the library itself produced for
    exactly this purpose (a BatchNorm-backward ``dres`` / ``dx`` or a dgrad output - marked ``_wsdl_fresh``; the mark
    survives the autograd engine, which either hands the tensor on untouched or sums a fan-out into the first
    arrival, still exclusively its own) and that are not views of something else.  Aliased gradients - the same tensor
    returned twice by an add node, channel slices of a concat's gradient, anything a user hook produced - fall back to
    an out-of-place add."""
    return (t is not None and getattr(t, "_wsdl_fresh", False) and t._base is None and t.is_contiguous()
            and t.dtype == torch.float32 and not t.requires_grad)


def affine_act_bwd(dy, y, scale, relu, want_dconv=True, want_dres=False):
    dy = _dense(dy, "dy")
    B, Cc, H, W = dy.shape
    dconv = torch.empty_like(dy) if want_dconv else None
    dres = torch.empty_like(dy) if want_dres else None
    amax = amax_slot(dy.device) if (want_dconv and CONV_ARITH[0] == 1) else None
    check(lib().wsdl_affine_act_bwd(_p(dy), _p(_dense(y) if relu else None), _p(scale), _p(dconv), _p(dres), B, Cc,
                                    H * W, int(relu), _p(amax), _stream()))
    if amax is not None:
        _publish_amax(dconv, amax)
    return dconv, dres


# ------------------------------------------------------------------------------------------ autograd
# Weight gradients are only consumed by the optimiser step, so they run on a side HIP stream: the MFMA-bound
# wgrad kernels overlap the HBM-bound BatchNorm backward / gradient-reduce kernels of the main chain.
OVERLAP_WGRAD = [True]
WGRAD_AFTER_DGRAD = [False]     # measured: 21.2 ms/step against 20.9 with the weight gradient enqueued first (profiles/r02_notes.md)
LAST_WGRAD_ON_MAIN = [os.environ.get("WSDL_LAST_WGRAD_ON_MAIN", "1") != "0"]   # the stem's weight gradient: see _wgrad_into
_side_streams = {}


def side_stream(device):
    if device.index is None:
        device = _norm_device(device)
    st = _side_streams.get(device)
    if st is None:
        st = _side_streams[device] = torch.cuda.Stream(device=device)
    return st


def join_side_stream(device):
    """Make the current stream wait for everything queued on the side stream (before the optimiser step / a
    gradient all-reduce reads the flat gradient buffer)."""
    st = _side_streams.get(_norm_device(device))
    if st is not None:
        stream_wait(raw_stream(device), st)


def _wgrad_into(param, x, dconv, wshape, stride, pad, dil, sink, x_amax=None, last=False, x_camax=None):
    """d(param) = wgrad(x, dconv) written straight into param.grad (a slice of the flat gradient buffer).
    ``last``: no input gradient follows (the network's first layer): nothing else is left for the main stream, so the
    kernel runs there, beside whatever the side stream still has queued, instead of behind it."""
    accumulate = not sink.take_fresh(param)
    split = _wgrad_split(wshape)
    x_amax = x_amax if x_amax is not None else amax_of(x, split)       # resolved on the MAIN stream (may launch a pass)
    dy_amax = amax_of(dconv, split)
    # per-channel maxima / pre-split rows travel as attributes of the tensors their producers returned
    dy_camax = getattr(dconv, "_wsdl_camax", None) if split else None
    dy_presplit = getattr(dconv, "_wsdl_presplit", None) if dy_camax is not None else None
    if not split:
        x_camax = None
    elif x_camax is None:
        x_camax = getattr(x, "_wsdl_camax", None)
    if x_camax is not None and x_camax.numel() != wshape[1]:
        x_camax = None                  # (a channel slice / concatenation of the tensor that carried it)
    if OVERLAP_WGRAD[0] and not (last and LAST_WGRAD_ON_MAIN[0]):
        # a densifying copy (never needed by the training step's own tensors) would be enqueued on the CURRENT stream: make it
        # happen before the side stream's wait, not inside conv2d_wgrad behind it
        x, dconv = _planes(x, "x")[0], _planes(dconv, "dy")[0]
        side = side_stream(x.device)
        hside = side.cuda_stream
        stream_wait(hside, _stream())               # dconv / x (and the zero_grad memset) are ready
        # launched ON the side stream by handle: torch's current stream stays the main one
        conv2d_wgrad(x, dconv, wshape, stride, pad, dil, out=param.grad, accumulate=accumulate, x_amax=x_amax,
                     dy_amax=dy_amax, stream=hside, defer=WGRAD_DEFER[0], x_camax=x_camax, dy_camax=dy_camax,
                     dy_presplit=dy_presplit)
        for t in (x_camax, dy_camax, dy_presplit):
            if t is not None:
                t.record_stream(side)
        x.record_stream(side)                       # keep the caching allocator from recycling them early
        dconv.record_stream(side)
    else:
        conv2d_wgrad(x, dconv, wshape, stride, pad, dil, out=param.grad, accumulate=accumulate, x_amax=x_amax,
                     dy_amax=dy_amax, defer=WGRAD_DEFER[0], x_camax=x_camax, dy_camax=dy_camax, dy_presplit=dy_presplit)
    sink.grad_ready(param)


def _sink_of(param):
    """Parameters owned by a FlatAdam carry ``_wsdl_grad_sink``: their gradient is written by the kernels
    straight into the optimiser's flat gradient buffer (no autograd accumulation pass, no extra add)."""
    sink = getattr(param, "_wsdl_grad_sink", None)
    if sink is None or param.grad is None or not param.grad.is_cuda:
        return None
    return sink


def _alias(t):
    """A tensor over the same memory that autograd knows nothing about (not a view of ``t`` in its books): what a kernel
    that writes into a slice of somebody else's buffer returns as its output."""
    return torch.empty(0, device=t.device, dtype=t.dtype).set_(t.untyped_storage(), t.storage_offset(), t.shape, t.stride())


class IdentityLink:
    """Shared by the two ``conv_bn_act`` calls that open and close an identity bottleneck (train mode).  The gradient of the
    block's input through the identity branch is [y > 0] * dy (dy: the gradient of the block's output, y its final ReLU).
    Routed through autograd it is a tensor the last BatchNorm backward writes (``dres``) and the first convolution's dgrad
    epilogue reads back.  With a link the last node leaves (dy, the ReLU's bits) here, returns no gradient for the residual
    input, and the first node - whose backward always runs later: its output's gradient depends on the rest of the block -
    adds [bits] * dy inside its dgrad epilogue, in dy's own buffer: one tensor write and one allocation less per block.
    Only when dy is a buffer the library owns (``_owned``); otherwise the ``dres`` path runs as before.  WSDL_IDENTITY_LINK=0
    switches it off (A/B; same result bit for bit)."""
    __slots__ = ("pending", "projection")

    def __init__(self, projection=False):
        # projection: the block's shortcut is a convolution + BatchNorm (the first block of a layer).  Then the link joins the
        # block's LAST node and that shortcut node: the last node hands the shortcut dy itself as "gradient" (an alias: no
        # tensor written) and the bits here; the shortcut's BatchNorm backward applies them (relu = 3) to what it reads.
        self.pending = None
        self.projection = projection


IDENTITY_LINK = [os.environ.get("WSDL_IDENTITY_LINK", "1") != "0"]


class _ConvBNAct(torch.autograd.Function):
    """Train-mode conv -> BatchNorm(batch statistics) -> (+residual) -> ReLU as one autograd node."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, residual, running_mean, running_var, stride, pad, dil, relu,
                momentum, eps, cache=None, passthrough=False, link=None, out_holder=None):
        # out_holder ([view into a larger buffer, shared amax slot] - a list, so that autograd does not see the tensors): the
        # BatchNorm kernel writes its output straight into a channel slice of a concatenation buffer (ASPP) - see concat_into
        # link (IdentityLink, shared by the first and the last node of an identity bottleneck): the last node's backward
        # does not write the masked gradient of the identity branch; it leaves (dy, mask bits) in the link and the first
        # node's dgrad epilogue adds [mask] * dy into dy's buffer - see IdentityLink
        ctx.link = link
        if link is not None:
            ctx.set_materialize_grads(False)
        # passthrough: also hand x back as a second output (the identity branch of a bottleneck).  The gradient that
        # arrives for it is then added inside the dgrad kernel's epilogue instead of by a separate autograd add.
        wf, wd = _cached_prep(cache, weight, x.requires_grad)
        x_amax = amax_of(x, _split_kc(x.shape[1], weight.shape[2] * weight.shape[3]) or _wgrad_split(weight.shape))
        conv = conv2d_fwd(x, wf, weight.shape, stride, pad, dil, x_amax=x_amax)
        y, mean, invstd, rbits = bn_train_fwd(conv, _dense(gamma), _dense(beta), running_mean, running_var, momentum, eps,
                                              residual, relu, out=_alias(out_holder[0]) if out_holder else None,
                                              want_mask=True, mask_if=bool(relu) and residual is not None,
                                              amax_into=out_holder[1] if out_holder else None)
        ctx.cfg = (stride, pad, dil, relu, tuple(weight.shape), tuple(x.shape), residual is not None)
        ctx.params = (weight, gamma, beta)
        ctx.x_amax = x_amax              # saved tensors come back as new Python objects: keep the scalar explicitly
        ctx.x_camax = getattr(x, "_wsdl_camax", None)     # ... and the per-channel maxima its producer published
        # the ReLU mask is recomputed from the conv output in the backward unless a residual was added: then the forward
        # kernel wrote it as bits (or, where it cannot - H*W not a multiple of 8 - y is kept and read)
        ctx.save_for_backward(x, conv, y if (relu and residual is not None and rbits is None) else None, gamma, mean, invstd,
                              wd, beta if relu else None, rbits)
        if passthrough:
            xv = x.view_as(x)
            if x_amax is not None:
                _publish_amax(xv, x_amax)
            if ctx.x_camax is not None:
                xv._wsdl_camax = ctx.x_camax
            return y, xv
        return y

    @staticmethod
    def backward(ctx, dy, dxres=None):
        x, conv, y, gamma, mean, invstd, wd, beta_s, rbits = ctx.saved_tensors
        stride, pad, dil, relu, wshape, xshape, has_res = ctx.cfg
        pw, pg, pb = ctx.params
        need_res = has_res and ctx.needs_input_grad[4]
        link = ctx.link
        if dy is None:
            raise WsdlError("conv -> BatchNorm node: no gradient arrived for its output")
        alias_dres = False
        if (link is not None and need_res and rbits is not None and IDENTITY_LINK[0] and tuple(dy.shape) == tuple(conv.shape)
                and dy.is_contiguous()):
            if link.projection:
                link.pending = (dy, rbits)   # consumed by the shortcut's node, which receives dy itself as its gradient
                need_res, alias_dres = False, True
            elif _owned(dy):
                link.pending = (dy, rbits)   # consumed by the block's first node (its backward runs after this one)
                need_res = False
        # the shortcut node of a projection block: its "gradient" is the block output's dy, to be masked by the bits
        shortcut_bits = None
        if link is not None and link.projection and not has_res:
            if link.pending is not None:
                src, shortcut_bits = link.pending
                link.pending = None
                if src.data_ptr() != dy.data_ptr() or tuple(src.shape) != tuple(dy.shape):
                    raise WsdlError("projection shortcut: the gradient that arrived is not the block output's (autograd "
                                    "copied or summed it) - set WSDL_IDENTITY_LINK=0")
        if shortcut_bits is not None:
            relu_b, y_b, beta_b, bits_b = True, None, None, shortcut_bits      # dy' = [bits] * dy, then this node's BatchNorm
        else:
            relu_b, y_b, beta_b, bits_b = relu, y, beta_s, rbits
        sg = _sink_of(pg) if (ctx.needs_input_grad[2] and ctx.needs_input_grad[3] and _sink_of(pg) is _sink_of(pb)) else None
        # the weight gradient's dY operand, written pre-split by the BatchNorm backward where that launch reads it so
        psb = wgrad_presplit_bytes(xshape, wshape, stride, pad, dil) if (ctx.needs_input_grad[1] and _sink_of(pw) is not None) else 0
        if sg is not None:
            fresh = sg.take_fresh(pg) & sg.take_fresh(pb)
            dconv, _, _, dres = bn_train_bwd(conv, dy, y_b, _dense(gamma), mean, invstd, relu_b, need_res,
                                             pg.grad, pb.grad, accumulate=not fresh, beta=beta_b, relu_mask=bits_b,
                                             presplit_bytes=psb)
            dgamma = dbeta = None
            sg.grad_ready(pg)
            sg.grad_ready(pb)
        else:
            dconv, dgamma, dbeta, dres = bn_train_bwd(conv, dy, y_b, _dense(gamma), mean, invstd, relu_b, need_res, beta=beta_b,
                                                      relu_mask=bits_b)
        if alias_dres:
            dres = dy
        pend = None
        if link is not None and not link.projection and link.pending is not None and not has_res:
            pend, link.pending = link.pending, None

        def input_grad():
            dx = None
            if ctx.needs_input_grad[0]:
                if wd is None:
                    raise WsdlError("conv backward: dgrad weights were not prepared")
                into, bits = None, None
                if pend is not None and tuple(pend[0].shape) == tuple(xshape):
                    into, bits = pend       # the block output's gradient + the final ReLU's bits: dx = dgrad(...) + [bits] * it
                elif _owned(dxres) and tuple(dxres.shape) == tuple(xshape):
                    into = dxres        # the identity branch's gradient (a fresh BN-backward output): dx = dgrad(...) + it
                dx = conv2d_dgrad(dconv, wd, wshape, xshape, stride, pad, dil, accumulate_into=into, acc_mask=bits)
                if into is None and dxres is not None:
                    dx = dx + dxres
                elif bits is not None and dxres is not None:
                    dx = dx + dxres          # (does not happen: the link replaces the gradient of the passthrough output)
            elif dxres is not None:
                dx = dxres
            return dx

        # Order of the two MFMA-bound kernels.  The weight gradient runs on the side stream behind everything enqueued on
        # the main stream so far.  WGRAD_AFTER_DGRAD: enqueue the input gradient FIRST, so the weight gradient starts when
        # it has finished and runs beside the NEXT layer's BatchNorm backward (HBM-bound) instead of beside this layer's
        # input gradient (two MFMA-bound kernels sharing one power budget gain nothing from overlapping).
        dx = input_grad() if WGRAD_AFTER_DGRAD[0] else None
        dw = None
        if ctx.needs_input_grad[1]:
            sw = _sink_of(pw)
            if sw is not None:
                _wgrad_into(pw, x, dconv, wshape, stride, pad, dil, sw, ctx.x_amax, last=not ctx.needs_input_grad[0],
                            x_camax=ctx.x_camax)
            else:
                dw = conv2d_wgrad(x, dconv, wshape, stride, pad, dil, x_amax=ctx.x_amax)
        if not WGRAD_AFTER_DGRAD[0]:
            dx = input_grad()
        return (dx, dw, dgamma if ctx.needs_input_grad[2] else None, dbeta if ctx.needs_input_grad[3] else None,
                dres, None, None, None, None, None, None, None, None, None, None, None, None)


ASPP_MULTI = [os.environ.get("WSDL_ASPP_MULTI", "1") != "0"]     # A/B: 0 = the branches as a chain of conv -> BatchNorm nodes
ASPP_GROUP_FWD = [os.environ.get("WSDL_ASPP_GROUP_FWD", "1") != "0"]   # A/B: 0 = the branches' forward convolutions one launch each


class _ConvBNBranches(torch.autograd.Function):
    """n train-mode conv -> BatchNorm -> ReLU branches over ONE input (the 1x1 and the three dilated 3x3 branches of ASPP), each
    writing its output into its channel slice of the concatenation buffer.  What it does differently from n chained
    ``_ConvBNAct`` nodes is the input gradient: ONE launch that accumulates all branches' taps per output tile
    (``conv2d_dgrad_multi``) instead of n launches that each read the running sum back and write it again.
    Arguments: x, momentum, eps, branches = [(stride-1 padding, dilation, cache, out_holder)], then per branch
    weight, gamma, beta, running_mean, running_var.  Returns y_0 .. y_{n-1}, x' (x routed through the node: the gradient
    arriving for it - the pooling branch's - is the launch's ``accumulate_into``)."""

    @staticmethod
    def forward(ctx, x, momentum, eps, branches, *tensors):
        n = len(branches)
        params = [tensors[5 * i:5 * i + 5] for i in range(n)]
        taps = max(w.shape[2] * w.shape[3] for (w, *_r) in params)
        x_amax = amax_of(x, _split_kc(x.shape[1], taps) or any(_wgrad_split(w.shape) for (w, *_r) in params))
        saved, outs, wshapes = [x], [], []
        preps = [_cached_prep(cache, w, x.requires_grad) for (_pad, _dil, cache, _h), (w, *_r) in zip(branches, params)]
        convs = None
        if ASPP_GROUP_FWD[0] and fwd_group_ok(n, tuple(x.shape), params[0][0].shape[0]):
            convs = conv2d_fwd_group(x, [pr[0] for pr in preps], [tuple(w.shape) for (w, *_r) in params],
                                     [b[1] for b in branches], x_amax=x_amax)
        for i, ((pad, dil, cache, holder), (w, gamma, beta, rm, rv)) in enumerate(zip(branches, params)):
            wf, wd = preps[i]
            conv = convs[i] if convs is not None else conv2d_fwd(x, wf, w.shape, 1, pad, dil, x_amax=x_amax)
            y, mean, invstd, _bits = bn_train_fwd(conv, _dense(gamma), _dense(beta), rm, rv, momentum, eps, None, True,
                                                  out=_alias(holder[0]), want_mask=True, mask_if=False, amax_into=holder[1])
            saved += [conv, gamma, mean, invstd, wd, beta]
            outs.append(y)
            wshapes.append(tuple(w.shape))
        ctx.cfg = ([(b[0], b[1]) for b in branches], wshapes, tuple(x.shape))
        ctx.params = [(w, g, b) for (w, g, b, _rm, _rv) in params]
        ctx.x_amax = x_amax
        ctx.x_camax = getattr(x, "_wsdl_camax", None)
        ctx.save_for_backward(*saved)
        xv = x.view_as(x)
        if x_amax is not None:
            _publish_amax(xv, x_amax)
        if ctx.x_camax is not None:
            xv._wsdl_camax = ctx.x_camax
        return (*outs, xv)

    @staticmethod
    def backward(ctx, *grads):
        geo, wshapes, xshape = ctx.cfg
        n = len(geo)
        dys, dxres = grads[:n], grads[n]
        saved = ctx.saved_tensors
        x = saved[0]
        dconvs, wds = [], []
        for i in range(n):
            conv, gamma, mean, invstd, wd, beta = saved[1 + 6 * i:7 + 6 * i]
            pw, pg, pb = ctx.params[i]
            if dys[i] is None:
                raise WsdlError("conv -> BatchNorm branches: no gradient arrived for branch %d" % i)
            sg = _sink_of(pg)
            if sg is None or sg is not _sink_of(pb) or _sink_of(pw) is None:
                raise WsdlError("conv -> BatchNorm branches: the parameters must be owned by a FlatAdam (gradient sinks)")
            fresh = sg.take_fresh(pg) & sg.take_fresh(pb)
            pad, dil = geo[i]
            dconv, _, _, _ = bn_train_bwd(conv, dys[i], None, _dense(gamma), mean, invstd, True, False, pg.grad, pb.grad,
                                          accumulate=not fresh, beta=beta,
                                          presplit_bytes=wgrad_presplit_bytes(xshape, wshapes[i], 1, pad, dil))
            sg.grad_ready(pg)
            sg.grad_ready(pb)
            _wgrad_into(pw, x, dconv, wshapes[i], 1, pad, dil, _sink_of(pw), ctx.x_amax, x_camax=ctx.x_camax)
            dconvs.append(dconv)
            wds.append(wd)
        dx = None
        if ctx.needs_input_grad[0]:
            into = dxres if (_owned(dxres) and tuple(dxres.shape) == tuple(xshape)) else None
            # the source that decides the column bands first: the smallest dilation among the 3x3 branches
            order = sorted(range(n), key=lambda i: (wshapes[i][2] == 1, geo[i][1]))
            dx = conv2d_dgrad_multi([dconvs[i] for i in order], [wds[i] for i in order], [wshapes[i] for i in order],
                                    [geo[i][1] for i in order], xshape, accumulate_into=into)
            if into is None and dxres is not None:
                dx = dx + dxres
        elif dxres is not None:
            dx = dxres
        return (dx, None, None, None) + (None,) * (5 * n)


def conv_bn_branches(x, branches, momentum, eps):
    """``branches``: [(conv module-like with .weight/.padding/.dilation and a cache dict, bn module-like, out_holder)] - see
    _ConvBNBranches; the caller (ASPP) has checked ``dgrad_multi_ok`` and train mode."""
    geo, tensors = [], []
    for conv, bn, holder in branches:
        geo.append((conv.padding, conv.dilation, conv.__dict__.setdefault("_wsdl_cache", {}), holder))
        tensors += [conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var]
    bump_stats_epoch()
    return _ConvBNBranches.apply(x, momentum, eps, geo, *tensors)


class _ConvAffineAct(torch.autograd.Function):
    """y = act(scale*conv(x,w) + shift + residual): eval-mode (folded) BN, conv bias, Linear."""

    @staticmethod
    def forward(ctx, x, weight, scale, shift, residual, stride, pad, dil, relu, shift_is_param, cache=None):
        need_dx = x.requires_grad
        wf, wd = _cached_prep(cache, weight, need_dx)
        # a folded-BatchNorm convolution (eval mode) feeds the next convolution directly: its epilogue publishes the amax
        y = conv2d_fwd(x, wf, weight.shape, stride, pad, dil, scale, shift, residual, relu,
                       want_amax=bool(relu) or scale is not None)
        ctx.cfg = (stride, pad, dil, relu, tuple(weight.shape), tuple(x.shape), residual is not None, shift_is_param)
        ctx.params = (weight, shift if shift_is_param else None)
        need_w = weight.requires_grad
        ctx.save_for_backward(x if need_w else None, y if relu else None, scale, wd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, scale, wd = ctx.saved_tensors
        stride, pad, dil, relu, wshape, xshape, has_res, shift_is_param = ctx.cfg
        need_res = has_res and ctx.needs_input_grad[4]
        dyc = _dense(dy, "dy")
        if relu or scale is not None or need_res:
            dconv, dres = affine_act_bwd(dyc, y, scale, relu, True, need_res)
            if need_res and not relu:
                dres = dyc
        else:
            dconv, dres = dyc, None
        pw, pbias = ctx.params
        dw = None
        if ctx.needs_input_grad[1]:
            sw = _sink_of(pw)
            if sw is not None and tuple(pw.grad.shape) == tuple(wshape):
                _wgrad_into(pw, x, dconv, wshape, stride, pad, dil, sw)
            else:
                dw = conv2d_wgrad(x, dconv, wshape, stride, pad, dil)
        dshift = None
        if ctx.needs_input_grad[3] and shift_is_param:
            # with scale None the shift is a plain bias: d/dshift = sum of the masked upstream gradient
            sb = _sink_of(pbias)
            if sb is not None:
                bias_grad(dconv, out=pbias.grad, accumulate=not sb.take_fresh(pbias))
                sb.grad_ready(pbias)
            else:
                dshift = bias_grad(dconv)
        dx = conv2d_dgrad(dconv, wd, wshape, xshape, stride, pad, dil) if ctx.needs_input_grad[0] else None
        return dx, dw, None, dshift, dres, None, None, None, None, None, None


class _BNTrain(torch.autograd.Function):
    """Stand-alone train-mode BatchNorm2d (batch statistics, running-statistics update): the same two kernels the
    fused conv -> BN node runs, without the convolution."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, momentum, eps):
        x = _dense(x, "x")
        y, mean, invstd = bn_train_fwd(x, _dense(gamma), _dense(beta), running_mean, running_var, momentum, eps)
        ctx.save_for_backward(x, gamma, mean, invstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, mean, invstd = ctx.saved_tensors
        dx, dgamma, dbeta, _ = bn_train_bwd(x, dy, None, _dense(gamma), mean, invstd, False, False)
        return dx, dgamma, dbeta, None, None, None, None


class _AffineAct(torch.autograd.Function):
    """y = act(scale[c]*x + shift[c]) on (B,C,H,W) [or (B,C)]: stand-alone eval-mode BatchNorm2d / ReLU."""

    @staticmethod
    def forward(ctx, x, scale, shift, relu):
        x = _dense(x, "x")
        B, Cc = (x.shape[0], x.shape[1]) if x.dim() >= 2 else (1, 1)
        HW = x.numel() // max(B * Cc, 1)
        y = torch.empty_like(x)
        check(lib().wsdl_affine_act_fwd(_p(x), _p(scale), _p(shift), _p(y), B, Cc, HW, int(relu), _stream()))
        ctx.relu, ctx.geom = relu, (B, Cc, HW)
        ctx.save_for_backward(y if relu else None, scale)
        return y

    @staticmethod
    def backward(ctx, dy):
        y, scale = ctx.saved_tensors
        B, Cc, HW = ctx.geom
        dy = _dense(dy, "dy")
        if not ctx.relu and scale is None:
            return dy, None, None, None
        dx = torch.empty_like(dy)
        check(lib().wsdl_affine_act_bwd(_p(dy), _p(y), _p(scale), _p(dx), _vp(0), B, Cc, HW, int(ctx.relu), _vp(0), _stream()))
        return dx, None, None, None


def batch_norm(x, gamma, beta, running_mean, running_var, momentum=0.1, eps=1e-5, training=True):
    """Stand-alone nn.BatchNorm2d.forward.  Eval mode folds the running statistics; like the fused eval path it
    treats gamma / beta as constants (the frozen classifier is the only eval-mode user of the reference)."""
    if training:
        bump_stats_epoch()
        return _BNTrain.apply(x, gamma, beta, running_mean, running_var, momentum, eps)
    scale, shift = bn_fold(gamma.detach(), beta.detach(), running_mean, running_var, eps)
    return _AffineAct.apply(x, scale, shift, False)


def relu(x):
    return _AffineAct.apply(x, None, None, True)


class _MaxPool3x3s2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _dense(x, "x")
        B, Cc, H, W = x.shape
        OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        y = torch.empty(B, Cc, OH, OW, device=x.device, dtype=torch.float32)
        am = torch.empty(B, Cc, OH, OW, device=x.device, dtype=torch.uint8)
        check(lib().wsdl_maxpool3x3s2_fwd(_p(x), _p(y), _p(am), B * Cc, H, W, _stream()))
        ctx.save_for_backward(am)
        ctx.xshape = tuple(x.shape)
        if getattr(x, "_wsdl_amax", None) is not None and getattr(x, "_wsdl_amax_version", x._version) == x._version:
            _publish_amax(y, x._wsdl_amax)       # max over windows of x: the input's bound holds (x >= 0 after ReLU or not)
        return y

    @staticmethod
    def backward(ctx, dy):
        (am,) = ctx.saved_tensors
        B, Cc, H, W = ctx.xshape
        dx = torch.empty(ctx.xshape, device=dy.device, dtype=torch.float32)
        check(lib().wsdl_maxpool3x3s2_bwd(_p(_dense(dy)), _p(am), _p(dx), B * Cc, H, W, _stream()))
        return dx


class _GlobalAvgPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _dense(x, "x")
        B, Cc, H, W = x.shape
        y = torch.empty(B, Cc, 1, 1, device=x.device, dtype=torch.float32)
        check(lib().wsdl_global_avgpool_fwd(_p(x), _p(y), B * Cc, H * W, _stream()))
        ctx.xshape = tuple(x.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, Cc, H, W = ctx.xshape
        dx = torch.empty(ctx.xshape, device=dy.device, dtype=torch.float32)
        check(lib().wsdl_global_avgpool_bwd(_p(_dense(dy)), _p(dx), B * Cc, H * W, 0, _stream()))
        dx._wsdl_fresh = True       # ASPP: the pooling branch's input gradient is summed into by the conv branches' dgrads
        return dx


class _Bilinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, H, W):
        x = _dense(x, "x")
        B, Cc, h, w = x.shape
        y = torch.empty(B, Cc, H, W, device=x.device, dtype=torch.float32)
        check(lib().wsdl_bilinear_fwd(_p(x), _p(y), B, Cc, h, w, H, W, 0, _stream()))
        ctx.shape = (B, Cc, h, w, H, W)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, Cc, h, w, H, W = ctx.shape
        dy, dy_bs = _planes(dy, "dy")
        dx = torch.empty(B, Cc, h, w, device=dy.device, dtype=torch.float32)
        check(lib().wsdl_bilinear_bwd(_p(dy), _p(dx), B, Cc, h, w, H, W, dy_bs, _stream()))
        return dx, None, None


class _Dropout(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, seed, mask, counter=None):
        x = _dense(x, "x")
        y = torch.empty_like(x)
        gen = mask is None
        if gen:
            mask = torch.empty(x.shape, device=x.device, dtype=torch.uint8)
        else:
            mask = _req(mask, "dropout mask", torch.uint8).contiguous()
        check(lib().wsdl_dropout_fwd(_p(x), _p(y), _p(mask), x.numel(), float(p), int(seed), int(gen), _p(counter),
                                     _stream()))
        if counter is not None and gen:
            add_int(counter, 1)         # on the stream, behind the kernel that read it (a node of a captured graph / a plan too)
        ctx.save_for_backward(mask)
        ctx.p = float(p)
        return y

    @staticmethod
    def backward(ctx, dy):
        (mask,) = ctx.saved_tensors
        dy = _dense(dy, "dy")
        dx = torch.empty_like(dy)
        check(lib().wsdl_dropout_bwd(_p(dy), _p(mask), _p(dx), dy.numel(), ctx.p, _stream()))
        return dx, None, None, None, None


class _ConcatChannels(torch.autograd.Function):
    """torch.cat(dim=1) with plane copies; backward hands out zero-copy channel slices."""

    @staticmethod
    def forward(ctx, *xs):
        B, _, H, W = xs[0].shape
        Cs = [int(t.shape[1]) for t in xs]
        out = torch.empty(B, sum(Cs), H, W, device=xs[0].device, dtype=torch.float32)
        off, tot = 0, sum(Cs) * H * W
        for t, c in zip(xs, Cs):
            t = _dense(t, "cat input")
            dst = out[:, off:off + c]
            check(lib().wsdl_copy_planes(_p(t), _p(dst), B, c, H * W, 0, tot, _stream()))
            off += c
        ctx.Cs = Cs
        return out

    @staticmethod
    def backward(ctx, dy):
        outs, off = [], 0
        for c in ctx.Cs:
            outs.append(dy[:, off:off + c])
            off += c
        return tuple(outs)


class _ConcatInto(torch.autograd.Function):
    """torch.cat(dim=1) into a buffer some of whose channel slices the producers have already written (``out_holder`` of
    ``conv_bn_act``): only the other inputs are copied.  ``holder`` = [buffer, shared amax slot or None]."""

    @staticmethod
    def forward(ctx, holder, *xs):
        base, slot = holder
        B, Ctot, H, W = base.shape
        Cs = [int(t.shape[1]) for t in xs]
        if sum(Cs) != Ctot:
            raise WsdlError("concat_into: channel counts do not add up to the buffer's")
        off, tot = 0, Ctot * H * W
        for t, c in zip(xs, Cs):
            dst = base[:, off:off + c]
            in_place = (t.data_ptr() == dst.data_ptr() and tuple(t.shape) == tuple(dst.shape) and t.stride() == dst.stride())
            if not in_place:
                t = _dense(t, "cat input")
                check(lib().wsdl_copy_planes(_p(t), _p(dst), B, c, H * W, 0, tot, _stream()))
                if slot is not None:      # this input's maximum joins the slot the in-place producers published into
                    check(lib().wsdl_amax(_p(t), B, c * H * W, c * H * W, _p(slot), 0, _stream()))
            off += c
        ctx.Cs = Cs
        out = _alias(base)
        if slot is not None:
            _publish_amax(out, slot)
        return out

    @staticmethod
    def backward(ctx, dy):
        outs, off = [None], 0
        for c in ctx.Cs:
            outs.append(dy[:, off:off + c])
            off += c
        return tuple(outs)


def concat_into(base, slot, xs):
    return _ConcatInto.apply([base, slot], *xs)


class _AddAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, relu):
        a, b = _dense(a, "a"), _dense(b, "b")
        y = torch.empty_like(a)
        check(lib().wsdl_add(_p(a), _p(b), _p(y), a.numel(), int(relu), _stream()))
        ctx.relu = relu
        ctx.save_for_backward(y if relu else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        if ctx.relu:
            dy4 = dy if dy.dim() == 4 else dy.reshape(dy.shape[0], -1, 1, 1)
            y4 = y.reshape(dy4.shape)
            g, _ = affine_act_bwd(dy4, y4, None, True, True, False)
            g = g.reshape(dy.shape)
        else:
            g = dy
        return g, g, None


class _SoftmaxCE(torch.autograd.Function):
    """nn.CrossEntropyLoss() on (B,C,H,W) logits, int64 labels; forward and gradient in one kernel.  Labels equal to
    ``ignore_index`` are left out of the mean (zero gradient); any other label outside [0, C) turns the loss into NaN
    (PyTorch raises there; see the kernel)."""

    @staticmethod
    def forward(ctx, logits, labels, ignore_index):
        logits = _dense(logits, "logits")
        labels = _req(labels, "labels", torch.int64).contiguous()
        B, Cc, H, W = logits.shape
        if tuple(labels.shape) != (B, H, W):
            raise WsdlError(f"cross entropy: labels {tuple(labels.shape)} do not match logits {tuple(logits.shape)}")
        loss = torch.empty((), device=logits.device, dtype=torch.float32)
        need = logits.requires_grad
        dl = torch.empty_like(logits) if need else None
        inv = torch.empty(1, device=logits.device, dtype=torch.float32)
        ws = workspace(lib().wsdl_reduce_workspace(), logits.device)
        check(lib().wsdl_softmax_ce_fwd_bwd(_p(logits), _p(labels), _p(loss), _p(dl), _p(inv), B, Cc, H, W, 1.0,
                                            int(ignore_index), _p(ws), ws.numel(), _stream()))
        ctx.save_for_backward(dl, inv)
        return loss

    @staticmethod
    def backward(ctx, g):
        dl, inv = ctx.saved_tensors
        out = torch.empty_like(dl)
        sc = torch.empty_like(inv)
        check(lib().wsdl_mul(_p(_dense(g.reshape(1))), _p(inv), _p(sc), 1, _stream()))      # upstream gradient x 1 / #pixels
        check(lib().wsdl_scale_by_device_scalar(_p(dl), _p(sc), _p(out), dl.numel(), _stream()))
        return out, None, None


class _PairwiseAffinityLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, preds, image, window, sigma_color, sigma_space, apply_softmax, normalise, cache=None):
        preds, image = _dense(preds, "preds"), _dense(image, "image")
        B, Cc, H, W = preds.shape
        if tuple(image.shape) != (B, 3, H, W):
            raise WsdlError(f"pairwise loss: image {tuple(image.shape)} does not match preds {tuple(preds.shape)}")
        loss = torch.empty(B if normalise else 1, device=preds.device, dtype=torch.float32)
        need = preds.requires_grad
        dp = torch.empty_like(preds) if need else None
        ws = workspace(lib().wsdl_pairwise_workspace(B, H, W), preds.device)
        check(lib().wsdl_pairwise_affinity_loss_fwd_bwd(_p(preds), _p(image), _p(loss), _p(dp), B, Cc, H, W,
                                                        int(window), float(sigma_color), float(sigma_space or 0.0),
                                                        int(apply_softmax), int(normalise), _p(cache), _p(ws), ws.numel(),
                                                        _stream()))
        ctx.save_for_backward(dp)
        ctx.normalise = normalise
        return loss if normalise else loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (dp,) = ctx.saved_tensors
        if ctx.normalise:
            # per-image upstream gradients: one scale launch per image (B is small on this path)
            out = torch.empty_like(dp)
            g = _dense(g.reshape(-1))
            n = dp[0].numel()
            for b in range(dp.shape[0]):
                check(lib().wsdl_scale_by_device_scalar(_p(dp[b]), _p(g[b:b + 1]), _p(out[b]), n, _stream()))
        else:
            out = torch.empty_like(dp)
            check(lib().wsdl_scale_by_device_scalar(_p(dp), _p(_dense(g.reshape(1))), _p(out), dp.numel(), _stream()))
        return out, None, None, None, None, None, None, None


# ------------------------------------------------------------------------------------------ functional API
def conv_bn_act(x, weight, gamma, beta, running_mean, running_var, stride, pad, dil, relu, residual=None,
                momentum=0.1, eps=1e-5, training=True, cache=None, passthrough=False, link=None, out_holder=None):
    """passthrough=True returns (y, x'): x' is x routed through this node, to be used as the identity branch so that
    its gradient is summed in the dgrad epilogue (train mode; eval mode returns x itself)."""
    if training:
        bump_stats_epoch()          # running statistics are about to be rewritten behind torch's back
        return _ConvBNAct.apply(x, weight, gamma, beta, residual, running_mean, running_var, stride, pad, dil,
                                bool(relu), momentum, eps, cache, bool(passthrough), link, out_holder)
    key = _cache_key(gamma, beta, running_mean, running_var) if cache is not None else None
    if cache is not None and cache.get("fold_key") == key:
        scale, shift = cache["fold"]
    else:
        scale, shift = bn_fold(gamma.detach(), beta.detach(), running_mean, running_var, eps)
        if cache is not None:
            cache["fold_key"], cache["fold"] = key, (scale, shift)
    y = _ConvAffineAct.apply(x, weight, scale, shift, residual, stride, pad, dil, bool(relu), False, cache)
    return (y, x) if passthrough else y


def conv_bias_act(x, weight, bias=None, stride=1, pad=0, dil=1, relu=False, residual=None, cache=None):
    return _ConvAffineAct.apply(x, weight, None, bias, residual, stride, pad, dil, bool(relu), bias is not None, cache)


def linear(x, weight, bias=None):
    """nn.Linear on (B,F) as a 1x1 convolution over a 1x1 map."""
    B, Fin = x.shape
    y = conv_bias_act(x.reshape(B, Fin, 1, 1), weight.reshape(weight.shape[0], Fin, 1, 1), bias)
    return y.reshape(B, -1)


def max_pool_3x3_s2(x):
    return _MaxPool3x3s2.apply(x)


def global_avg_pool(x):
    return _GlobalAvgPool.apply(x)


def class_logit_head(h, weight, bias=None, class_idx=None):
    """LayerCAM's class-logit head on layer4's output h (B,C,H,W): logits = fc(avgpool(h)), the class per image (class_idx or the
    arg-max) and d logit[class] / d h = weight[class] / HW per pixel - wsdl_class_logit_head; fc's own parameter gradients are
    not formed.  Returns (logits (B,K), cls (B,) int32, dh like h)."""
    h = _dense(h, "h")
    B, Cc, H, W = h.shape
    K = weight.shape[0]
    if tuple(weight.shape) != (K, Cc) or not weight.is_contiguous() or weight.dtype != torch.float32:
        raise WsdlError("class_logit_head: weight must be a contiguous float32 (classes, channels) tensor")
    pooled = torch.empty(B, Cc, device=h.device, dtype=torch.float32)
    check(lib().wsdl_global_avgpool_fwd(_p(h), _p(pooled), B * Cc, H * W, _stream()))
    logits = torch.empty(B, K, device=h.device, dtype=torch.float32)
    cls = torch.empty(B, device=h.device, dtype=torch.int32)
    dh = torch.empty_like(h)
    if class_idx is not None:
        class_idx = class_idx.to(device=h.device, dtype=torch.int64).contiguous().view(-1)
        if class_idx.numel() != B:
            raise WsdlError("class_logit_head: one class index per image")
    check(lib().wsdl_class_logit_head(_p(pooled), _p(weight), _p(bias) if bias is not None else None,
                                      _p(class_idx) if class_idx is not None else None, _p(logits), _p(cls), _p(dh), B, Cc, K, H * W,
                                      _stream()))
    return logits, cls, dh


def bilinear_resize(x, size):
    return _Bilinear.apply(x, int(size[0]), int(size[1]))


DROPOUT_SEED_OFFSET = [0]      # dp.init_distributed: a different offset on every rank, so replicas draw different masks


def dropout(x, p, training=True, seed=None, mask=None, counter=None):
    """``counter``: a device int64 call counter (nn.Dropout keeps one): the mask then depends on (seed, counter) and the
    module draws its host seed only once - required for hipGraph replay, where a per-call host seed would be frozen."""
    if not training or p == 0.0:
        return x
    if seed is None:
        seed = (int(torch.randint(0, 2 ** 62, (1,)).item()) + DROPOUT_SEED_OFFSET[0]) % (1 << 62)
    return _Dropout.apply(x, p, seed, mask, counter)


def concat_channels(xs):
    return _ConcatChannels.apply(*xs)


def add_act(a, b, relu=False):
    return _AddAct.apply(a, b, bool(relu))


def cross_entropy(logits, labels, ignore_index=-100):
    return _SoftmaxCE.apply(logits, labels, ignore_index)


class _LovaszSoftmax(torch.autograd.Function):
    """lovasz_softmax(probas, labels, classes, per_image=False, ignore) - reference
    TraditionalModel/LossFunctions/Lovasz-Softmax_Loss.py:146-192; loss and d loss / d probas in one call (a stable
    radix sort, a scan and one pass per class on the device)."""

    @staticmethod
    def forward(ctx, probas, labels, classes_all, ignore):
        probas = _dense(probas, "probas")
        labels = _req(labels, "labels", torch.int64).contiguous()
        B, Cc, H, W = probas.shape
        if tuple(labels.shape) != (B, H, W):
            raise WsdlError(f"lovasz_softmax: labels {tuple(labels.shape)} do not match probas {tuple(probas.shape)}")
        loss = torch.empty((), device=probas.device, dtype=torch.float32)
        dp = torch.empty_like(probas) if probas.requires_grad else None
        ws = workspace(lib().wsdl_lovasz_softmax_workspace(B, Cc, H, W), probas.device)
        check(lib().wsdl_lovasz_softmax_fwd_bwd(_p(probas), _p(labels), _p(loss), _p(dp), B, Cc, H, W, int(classes_all),
                                                int(ignore), _p(ws), ws.numel(), _stream()))
        ctx.save_for_backward(dp)
        return loss

    @staticmethod
    def backward(ctx, g):
        (dp,) = ctx.saved_tensors
        return dp * g, None, None, None


def lovasz_softmax(probas, labels, classes="present", per_image=False, ignore=None):
    """probas (B,C,H,W) class probabilities (or (B,H,W): one sigmoid map), labels (B,H,W)."""
    if classes not in ("present", "all"):
        raise WsdlError("lovasz_softmax: classes must be 'present' or 'all' (an explicit class list is not supported)")
    if probas.dim() == 3:
        probas = probas.unsqueeze(1)
    ign = -(1 << 62) if ignore is None else int(ignore)
    if per_image:
        vals = [_LovaszSoftmax.apply(probas[b:b + 1], labels[b:b + 1], classes == "all", ign) for b in range(probas.shape[0])]
        return torch.stack(vals).mean()
    return _LovaszSoftmax.apply(probas, labels, classes == "all", ign)


def pairwise_affinity_loss(preds, image, window=5, sigma_color=0.1, sigma_space=0.0, apply_softmax=True,
                           normalise=0, cache=None):
    """``cache``: ``pairwise_cache(image, window, sigma_color)`` when the image stays fixed over many evaluations."""
    return _PairwiseAffinityLoss.apply(preds, image, window, sigma_color, sigma_space, bool(apply_softmax),
                                       int(normalise), cache)


def pairwise_cache(image, window=5, sigma_color=0.1):
    """The image's colour affinities for the forward half of the window ((window^2-1)/2, B, H, W)."""
    image = _dense(image, "image")
    B, _, H, W = image.shape
    out = torch.empty((window * window - 1) // 2, B, H, W, device=image.device, dtype=torch.float32)
    check(lib().wsdl_pairwise_cache(_p(image), _p(out), B, H, W, int(window), float(sigma_color), _stream()))
    return out


def compute_affinities(image, sigma_color=0.1, sigma_space=5, window_size=5):
    image = _dense(image, "image")
    B, _, H, W = image.shape
    K = window_size * window_size - 1
    out = torch.empty(K, B, 1, H, W, device=image.device, dtype=torch.float32)
    check(lib().wsdl_compute_affinities(_p(image), _p(out), B, H, W, int(window_size), float(sigma_color),
                                        float(sigma_space or 0.0), _stream()))
    return out


def layercam_epilogue(acts, grads, out_hw=(224, 224), alpha=1.0, variant="modular", thresh=None):
    """acts/grads: lists of (B,C,h,w) device tensors -> cam (B,outH,outW) [, uint8 mask]."""
    n = len(acts)
    acts = [_dense(a.detach(), "act") for a in acts]
    grads = [_dense(g.detach(), "grad") for g in grads]
    B = acts[0].shape[0]
    for a, g in zip(acts, grads):
        if a.shape != g.shape or a.shape[0] != B:
            raise WsdlError("layercam: activation / gradient shape mismatch")
    IA = C.c_int * n
    Cs, hs, ws_ = IA(*[a.shape[1] for a in acts]), IA(*[a.shape[2] for a in acts]), IA(*[a.shape[3] for a in acts])
    PA = _vp * n
    pa, pg = PA(*[_p(a) for a in acts]), PA(*[_p(g) for g in grads])
    dev = acts[0].device
    cam = torch.empty(B, out_hw[0], out_hw[1], device=dev, dtype=torch.float32)
    mask = torch.empty(B, out_hw[0], out_hw[1], device=dev, dtype=torch.uint8) if thresh is not None else None
    nbytes = lib().wsdl_layercam_workspace(n, B, Cs, hs, ws_)
    if nbytes == 0:
        raise WsdlError("layercam: bad layer geometry")
    ws = workspace(nbytes, dev)
    var = {"modular": 0, "notebook": 1}[variant]
    check(lib().wsdl_layercam_epilogue(pa, pg, Cs, hs, ws_, n, B, out_hw[0], out_hw[1], float(alpha), var, _p(cam),
                                       float(thresh) if thresh is not None else -1.0, _p(mask), _p(ws), ws.numel(),
                                       _stream()))
    return (cam, mask) if thresh is not None else cam


def keep_largest_batched(masks):
    """(N,H,W) (or (H,W)) uint8 / bool device masks -> uint8 {0,1}: the largest 8-connected component of each mask, the
    first in raster order on area ties, an empty mask stays empty (reference PsuedoMasks.py:15-21, per image on the host
    there).  No host synchronisation."""
    if masks.dtype == torch.bool:
        masks = masks.to(torch.uint8)
    if masks.dtype != torch.uint8:
        raise WsdlError("keep_largest: masks must be uint8 or bool")
    m = _req(masks, "masks", torch.uint8).contiguous()
    m3 = m.view(-1, m.shape[-2], m.shape[-1])
    out = torch.empty_like(m3)
    if m3.numel() == 0:
        return out.view(m.shape)
    n, h, w = m3.shape
    nbytes = lib().wsdl_keep_largest_workspace(n, h, w)
    if nbytes == 0:
        raise WsdlError("keep_largest: bad mask geometry")
    ws = workspace(nbytes, m.device)
    check(lib().wsdl_keep_largest(_p(m3), _p(out), n, h, w, _p(ws), ws.numel(), _stream()))
    return out.view(m.shape)


def plane_relu_minmax(x):
    """(..., h, w) -> per-plane (relu(x) - min) / (max + 1e-8)."""
    x = _dense(x, "x")
    hw = x.shape[-1] * x.shape[-2]
    y = torch.empty_like(x)
    check(lib().wsdl_plane_relu_minmax(_p(x), _p(y), x.numel() // hw, hw, _stream()))
    return y


def adam_step_flat(p, g, m, v, lr, beta1, beta2, eps, step, grad_scale=1.0, step_dev=None, hyper_dev=None):
    """``step_dev``: device int32 step number (used instead of ``step`` - hipGraph replay).  ``hyper_dev``: device floats
    (lr, beta1, beta2, eps, grad_scale) used instead of the host values (with ``step_dev``): nothing of the launch changes
    when a schedule changes them."""
    for t in (p, g, m, v):
        _req(t, "adam buffer")
    if hyper_dev is not None and step_dev is not None:
        check(lib().wsdl_adam_step_dev(_p(p), _p(g), _p(m), _p(v), p.numel(), _p(hyper_dev), _p(step_dev), _stream()))
        return
    check(lib().wsdl_adam_step(_p(p), _p(g), _p(m), _p(v), p.numel(), float(lr), float(beta1), float(beta2),
                               float(eps), int(step), _p(step_dev), float(grad_scale), _stream()))


def softmax_channels(x):
    return _SoftmaxChannels.apply(x)


class _SoftmaxChannels(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _dense(x, "x")
        B, Cc = x.shape[0], x.shape[1]
        HW = x.numel() // (B * Cc)
        y = torch.empty_like(x)
        check(lib().wsdl_softmax_fwd(_p(x), _p(y), B, Cc, HW, _stream()))
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dy = _dense(dy, "dy")
        B, Cc = y.shape[0], y.shape[1]
        dx = torch.empty_like(y)
        check(lib().wsdl_softmax_bwd(_p(y), _p(dy), _p(dx), B, Cc, y.numel() // (B * Cc), _stream()))
        return dx


class _KLDivBatchMean(torch.autograd.Function):
    """F.kl_div((xn + 1e-8).log(), s, reduction='batchmean') and d/dxn."""

    @staticmethod
    def forward(ctx, xn, s):
        xn, s = _dense(xn, "xn"), _dense(s, "s")
        loss = torch.empty((), device=xn.device, dtype=torch.float32)
        dxn = torch.empty_like(xn)
        ws = workspace(lib().wsdl_reduce_workspace(), xn.device)
        check(lib().wsdl_kl_div_fwd_bwd(_p(xn), _p(s), _p(loss), _p(dxn), xn.numel(), xn.shape[0], _p(ws), ws.numel(),
                                        _stream()))
        ctx.save_for_backward(dxn)
        return loss

    @staticmethod
    def backward(ctx, g):
        (dxn,) = ctx.saved_tensors
        out = torch.empty_like(dxn)
        check(lib().wsdl_scale_by_device_scalar(_p(dxn), _p(_dense(g.reshape(1))), _p(out), dxn.numel(), _stream()))
        return out, None


class _ScaleMean(torch.autograd.Function):
    """w * mean(x) of a device scalar / vector as launches of the library (a launch plan sees them; ``0.1 * loss`` and
    ``.mean()`` are kernels of the tensor library)."""

    @staticmethod
    def forward(ctx, x, w):
        xs = _dense(x, "x")
        out = torch.empty((), device=x.device, dtype=torch.float32)
        check(lib().wsdl_scale_mean(_p(xs), xs.numel(), float(w), _p(out), _stream()))
        ctx.w, ctx.shape = float(w), tuple(x.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        n = 1
        for d in ctx.shape:
            n *= d
        dx = torch.empty(ctx.shape, device=g.device, dtype=torch.float32)
        check(lib().wsdl_scale_fill(_p(_dense(g.reshape(1))), ctx.w / n, _p(dx), n, _stream()))
        return dx, None


def scale_mean(x, w=1.0):
    """``w * x.mean()`` (``w * x`` for a scalar) -> 0-dim tensor."""
    return _ScaleMean.apply(x, float(w))


class _Fanout(torch.autograd.Function):
    """n handles on one tensor whose gradients are summed by the library's add kernel, in the order of the handles.  A tensor
    consumed twice has its gradients added by the autograd engine with a kernel of the tensor library - which a launch plan
    does not see (the logits feeding the cross-entropy AND a pairwise loss)."""

    @staticmethod
    def forward(ctx, x, n):
        ctx.set_materialize_grads(False)
        outs = tuple(_alias(x) for _ in range(n))
        a = getattr(x, "_wsdl_amax", None)
        if a is not None and getattr(x, "_wsdl_amax_version", x._version) == x._version:
            for o in outs:
                _publish_amax(o, a)
        return outs

    @staticmethod
    def backward(ctx, *gs):
        gs = [_dense(g, "gradient") for g in gs if g is not None]
        if not gs:
            return None, None
        if len(gs) == 1:
            return gs[0], None
        acc = torch.empty_like(gs[0])
        check(lib().wsdl_add(_p(gs[0]), _p(gs[1]), _p(acc), acc.numel(), 0, _stream()))
        for g in gs[2:]:
            check(lib().wsdl_add(_p(acc), _p(g), _p(acc), acc.numel(), 0, _stream()))
        return acc, None


def fanout(x, n=2):
    """``n`` handles on ``x`` for ``n`` consumers (see _Fanout); without autograd just ``x`` n times."""
    if not (torch.is_grad_enabled() and x.requires_grad and x.is_cuda):
        return (x,) * n
    return _Fanout.apply(x, int(n))


def add_scalars(a, b):
    """a + b for two 0-dim device tensors through the library's add kernel."""
    return _AddAct.apply(a.reshape(1), b.reshape(1), False).reshape(())


def kl_div_batchmean(xn, s):
    return _KLDivBatchMean.apply(xn, s)


# ---- profiler ranges (roctx): WSDL_ROCTX=1 in the environment (or ranges_enable(True)) brackets the phases of a training step
# and of a CAM batch, and every instrumented kernel class, with named ranges for `rocprofv3 --marker-trace`
RANGES = [False]


def ranges_enable(on=True):
    check(lib().wsdl_range_enable(int(bool(on))))
    RANGES[0] = bool(on)


class prof_range:
    """``with ops.prof_range("backward"): ...`` - a named roctx range when ranges are enabled, nothing otherwise."""
    __slots__ = ("name",)

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        if RANGES[0]:
            lib().wsdl_range_push(self.name.encode())

    def __exit__(self, *exc):
        if RANGES[0]:
            lib().wsdl_range_pop()
        return False


if os.environ.get("WSDL_ROCTX") == "1":
    try:
        ranges_enable(True)
    except Exception:           # the profiler SDK is not there: run without ranges
        RANGES[0] = False


PROF_ON = [False]      # per-launch event brackets are being recorded (bench.py's roofline pass): a plan replay has none


def prof_enable(on):
    check(lib().wsdl_prof_enable(int(on)))
    PROF_ON[0] = bool(on)


def prof_reset():
    check(lib().wsdl_prof_reset())


PROF_NCLASSES = 23


def prof_class_name(cls):
    return lib().wsdl_prof_class_name(int(cls)).decode()


def prof_collect(cls):
    n, ms, work, exe, byt = C.c_longlong(0), C.c_double(0), C.c_double(0), C.c_double(0), C.c_double(0)
    check(lib().wsdl_prof_collect(int(cls), C.byref(n), C.byref(ms), C.byref(work), C.byref(exe), C.byref(byt)))
    return n.value, ms.value, work.value, exe.value, byt.value
