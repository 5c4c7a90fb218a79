"""hipGraph replay of the training step.

One training iteration is ~600 kernel launches issued from Python through ctypes (13-17 ms of host time per step
against ~21 ms of GPU time at B=16, 256x256): the GPU stays ahead only while a host core is free for it, and every
launch leaves a few microseconds of gap on the stream.  ``GraphedTrainStep`` captures forward + loss + backward + Adam +
next step's weight re-layout ONCE per (shape, model) into a HIP graph (``torch.cuda.CUDAGraph`` on ROCm is a hipGraph)
and replays it: one host call per step.

What makes the step capturable (everything that used to be a host value per step now lives on the device):
  * Adam's step number           -> ``FlatAdam.step_dev`` (bias corrections computed in the kernel),
  * Adam's lr / betas / eps / grad_scale -> ``FlatAdam.hyper_dev`` (five device floats the kernel reads; the wrapper copies
                                    changed values there before the capture and before every replay: a schedule needs no new capture),
  * dropout masks                -> per-module device call counters mixed into a seed drawn once (``nn.Dropout``),
  * the convolutions' amax slots -> a pool allocated (and therefore re-zeroed) inside the captured region,
  * weight layouts               -> re-written IN PLACE at the end of the step, the main stream joins the side stream
                                    instead of handing events across steps.
Host-side bookkeeping that replay skips (``FlatAdam.step_count``, BatchNorm ``num_batches_tracked``, the cache epochs)
is advanced by the wrapper.  Data parallelism stays eager: the bucket reducer's control-plane exchange is host code.

The first ``warmup`` calls run eagerly (they are real training steps: allocator, lazy library state and the
reducer-free path settle), the next call captures and replays; the arithmetic is the eager path's, kernel for kernel.
"""
import torch

from . import nn as wnn
from . import ops


class GraphedTrainStep:
    def __init__(self, model, optimizer, extra_loss=None, warmup=2):
        from .TraditionalModel.SegmentationModel import train_step
        self._train_step = train_step
        self.model, self.opt, self.extra_loss = model, optimizer, extra_loss
        self.warmup = max(1, int(warmup))
        self.calls = 0
        self.graph = None
        self.key = None
        self._bns = [m for m in model.modules() if isinstance(m, wnn.BatchNorm2d)]
        self._convs = [m for m in model.modules() if isinstance(m, wnn.Conv2d) and m.weight.requires_grad]

    def _eager(self, images, masks):
        return self._train_step(self.model, self.opt, images, masks, self.extra_loss)

    def _capture(self, images, masks):
        dev = images.device
        self.s_images, self.s_masks = images.clone(), masks.clone()
        opt = self.opt
        saved_hook = opt.post_step_hook
        # weight layouts: end-of-step re-layout in place and without events (the captured step joins the side stream;
        # a replay must find the layouts at the addresses the captured forward reads)
        segmented = getattr(opt, "segments", None) is not None
        if segmented:
            opt.capture_mode = True
        opt.post_step_hook = lambda: ops.prefetch_weight_layouts(self._convs, use_events=False)
        torch.cuda.synchronize(dev)
        for m in self._convs:                   # no waits on events recorded outside the capture
            m.__dict__.setdefault("_wsdl_cache", {})["prep_event"] = None
        ops.reset_amax_pool(dev)                # the first slot request inside the capture allocates + zeroes a pool there
        g = torch.cuda.CUDAGraph()
        count0, pend0 = opt.step_count, [b._pending_steps for b in self._bns]
        try:
            with torch.cuda.graph(g):
                self.s_loss = self._eager(self.s_images, self.s_masks)
                ops.join_side_stream(dev)
        finally:
            opt.post_step_hook = saved_hook
            if segmented:
                opt.capture_mode = False
            ops.reset_amax_pool(dev)            # eager code must not hand out the graph's slots
        # capturing executed the Python side of one step but no kernel: undo its host bookkeeping (replay redoes it)
        opt.step_count = count0
        for b, p0 in zip(self._bns, pend0):
            b._pending_steps = p0
        self.graph = g

    def __call__(self, images, masks):
        """One training iteration on (images, masks); returns the (device) loss, like ``train_step``."""
        self.calls += 1
        # the Adam kernel reads lr / betas / eps / grad_scale from device memory (FlatAdam.hyper_dev): a scheduler, a manual
        # decay or a data-parallel reducer changing one of them changes memory, not the captured node - no new capture, but the
        # copy has to happen OUTSIDE the capture and before each replay (sync_hyper refuses to run inside one)
        opt = self.opt
        key = (tuple(images.shape), tuple(masks.shape), images.device, self.model.training)
        if self.calls <= self.warmup or not self.model.training:
            return self._eager(images, masks)
        opt.sync_hyper()
        if self.graph is None or key != self.key:
            self.key = key
            self._capture(images, masks)
        if images.data_ptr() != self.s_images.data_ptr():
            self.s_images.copy_(images)
        if masks.data_ptr() != self.s_masks.data_ptr():
            self.s_masks.copy_(masks)
        self.graph.replay()
        # host-side twins of what the replayed kernels did
        self.opt.step_count += 1
        for b in self._bns:
            if b.training:
                b._pending_steps += 1
        ops.bump_param_epoch()
        ops.bump_stats_epoch()
        return self.s_loss
