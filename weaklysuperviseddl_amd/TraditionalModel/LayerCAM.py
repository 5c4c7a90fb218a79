"""LayerCAMGenerator on the HIP path.

Mirrors reference TraditionalModel/LayerCAM.py:7-81 (``variant="modular"``, the default) and the
notebook twin TraditionalModel/AlternatingDirectionCutLoss.py:216-293 (``variant="notebook"``;
SURVEY.md D3: extra ``**alpha`` + second min-max per layer, no final power).

What runs where:
  * the classifier forward and the class-logit backward are the HIP conv / pooling kernels behind
    ``FrozenResNetCAM`` (eval-mode BN folded into the conv epilogue);
  * ReLU(grad*act) channel sum, ReLU, per-image min-max, bilinear up-sample to 224x224, layer mean,
    clamp/pow (and optionally the PsuedoMasks threshold) are ``wsdl_layercam_epilogue``.

``generate(images, alpha=1.0, class_idx=None)`` accepts both reference keyword orders (D2) and
``__call__`` is an alias.  Two ways to obtain (activation, gradient) pairs:
  * hook path (any module exposing the named layers; what the reference does - the backward runs all
    the way to the image);
  * staged path (default when the model is our FrozenResNetCAM and the targets are among
    layer3/layer4): layers 0..3 run without a graph and the backward stops at the earliest target's
    output - the only part of the backward LayerCAM consumes.  Same numbers, ~half the work.
``generate_batch`` does B images at once (eval-mode BN makes images independent), removing the
reference's per-image host round trips (``torch.cuda.empty_cache()`` LayerCAM.py:79).
"""
import os

import torch

from .. import ops
from .ClassificationModel import FrozenResNetCAM

_ORDER = ["layer0", "layer1", "layer2", "layer3", "layer4"]


class LayerCAMGenerator:
    def __init__(self, model, target_layer_names=("layer3", "layer4"), variant="modular", out_hw=(224, 224),
                 staged=None, auto_graph=True, fc_param_grads=False):
        self.model = model.eval()
        # staged path: False = the class-logit head as ONE library call (ops.class_logit_head: logits, class choice and the
        # gradient seed W[class] / HW) - fc.weight.grad / fc.bias.grad, which the reference's backward accumulates as a side
        # effect and nothing reads, are not formed; True = fc and the pool as autograd nodes, parameter gradients included
        self.fc_param_grads = bool(fc_param_grads) or os.environ.get("WSDL_CAM_FUSED_HEAD", "1") == "0"
        # staged path only: a repeated call (same shapes, same model state) replays a hipGraph of the batch (_lane_batch)
        self.auto_graph = bool(auto_graph) and os.environ.get("WSDL_CAM_GRAPH", "1") != "0" and \
            os.environ.get("WSDL_CAM_SELF_GRAPH", "1") != "0"
        self.target_layer_names = list(target_layer_names)
        self.variant = variant
        self.out_hw = tuple(out_hw)
        self.activations, self.gradients = {}, {}
        if staged is None:
            staged = isinstance(model, FrozenResNetCAM) and all(n in _ORDER[1:] for n in self.target_layer_names)
        self.staged = staged
        self._hooks = []
        if not staged:
            for name in self.target_layer_names:
                layer = getattr(self.model, name)
                self._hooks.append(layer.register_forward_hook(self._fwd(name)))
                self._hooks.append(layer.register_full_backward_hook(self._bwd(name)))

    def _fwd(self, name):
        def hook(_m, _inp, out):
            self.activations[name] = out
        return hook

    def _bwd(self, name):
        def hook(_m, _gin, gout):
            self.gradients[name] = gout[0]
        return hook

    # -- (activation, gradient) capture ----------------------------------------------------------
    def _capture_hooks(self, x, class_idx):
        x = x.detach().clone().requires_grad_()
        with torch.enable_grad():
            logits, _ = self.model(x)
            if class_idx is None:
                class_idx = logits.argmax(dim=1)
            score = logits.gather(1, class_idx.view(-1, 1)).squeeze()
            score.backward(torch.ones_like(score))
        return logits

    def _capture_staged(self, x, class_idx):
        m = self.model
        first = min(_ORDER.index(n) for n in self.target_layer_names)
        feats = {}
        with torch.no_grad():
            h = x
            for name in _ORDER[:first + 1]:
                h = getattr(m, name)(h)
                feats[name] = h
        with torch.enable_grad():
            h = h.detach().requires_grad_()
            feats[_ORDER[first]] = h
            for name in _ORDER[first + 1:]:
                h = getattr(m, name)(h)
                if name in self.target_layer_names:
                    h.retain_grad()
                feats[name] = h
            if self.fc_param_grads or not h.is_cuda:
                logits = m.fc(m.avgpool(h).flatten(1))
                if class_idx is None:
                    class_idx = logits.argmax(dim=1)
                score = logits.gather(1, class_idx.view(-1, 1)).squeeze()
                score.backward(torch.ones_like(score))
            else:
                # ~20 launches of the tail (pool, fc on the matrix cores, gather, scatter, fc's weight / bias / input gradients,
                # the pool's backward) as three: 160 -> ~25 us of a 2.6 ms batch of 8
                logits, _cls, dh = ops.class_logit_head(h.detach(), m.fc.weight.detach(), m.fc.bias.detach(), class_idx)
                h.backward(dh)
        for n in self.target_layer_names:
            self.activations[n] = feats[n]
            self.gradients[n] = feats[n].grad
        return logits

    # -- public API ---------------------------------------------------------------------------------
    def _state_key(self):
        """What a captured batch depends on besides its inputs: THIS model's parameters and statistics (a replay reads the
        weight layouts and folded BatchNorm constants of capture time); any change re-captures.  Per-model indicators, not
        the library's global epochs (another model's optimiser step must not invalidate this graph): tensor versions (torch-side
        writes), the step counts of the optimisers that own parameters of the model (FlatAdam writes through raw pointers)
        and the BatchNorm modules' train-mode forward counters (running statistics written by the kernels)."""
        from .. import nn as wnn
        m = self.model
        versions = sum(t._version for t in m.parameters()) + sum(t._version for t in m.buffers())
        sinks = {id(k): k.step_count for k in (getattr(p, "_wsdl_grad_sink", None) for p in m.parameters()) if k is not None}
        bn_steps = sum(b._pending_steps for b in m.modules() if isinstance(b, wnn.BatchNorm2d))
        return (ops.LAYOUT_EPOCH[0], m.training, versions, tuple(sorted(sinks.items())), bn_steps)

    def generate_batch(self, images, alpha=1.0, class_idx=None, thresh=None):
        """images (B,3,H,W), class_idx LongTensor (B,) or None -> cam (B,outH,outW) [, uint8 mask].
        Staged path on the device: the second call with the same shapes captures the batch (~130 launches that cost the host
        as long to issue as the GPU needs to run them) into a hipGraph, later calls replay it - same kernels, same results."""
        if self.auto_graph and self.staged and images.is_cuda and not torch.cuda.is_current_stream_capturing():
            # one batch = lane 0 of generate_batches: its graph, on its stream (a replay launched into the caller's stream -
            # torch's default stream - took anything from 0.35 to 0.53 ms/img after a training step had run in the process;
            # on the lane's stream it is the 0.345 of the isolated measurement)
            dev = images.device
            lane = self._get_lanes(1, dev)[0]
            st, cur = lane["stream"], torch.cuda.current_stream(dev)
            if class_idx is not None:
                class_idx = class_idx.to(dev).view(-1)
                class_idx.record_stream(st)
            st.wait_stream(cur)
            images.record_stream(st)
            with torch.cuda.stream(st):
                out = self._lane_batch(lane, images, float(alpha), class_idx, thresh, self._state_key())
            for t in (out if isinstance(out, tuple) else (out,)):
                t.record_stream(cur)
            cur.wait_stream(st)
            self.activations, self.gradients = lane["gen"].activations, lane["gen"].gradients
            return out
        return self._generate_batch_eager(images, alpha, class_idx, thresh)

    def _generate_batch_eager(self, images, alpha=1.0, class_idx=None, thresh=None):
        self.activations.clear()
        self.gradients.clear()
        if class_idx is not None:
            class_idx = class_idx.to(images.device).view(-1)
        with ops.prof_range("layercam/forward+class-logit backward"):
            (self._capture_staged if self.staged else self._capture_hooks)(images, class_idx)
        acts = [self.activations[n].detach() for n in self.target_layer_names]
        grads = [self.gradients[n].detach() for n in self.target_layer_names]
        with ops.prof_range("layercam/epilogue"):
            return ops.layercam_epilogue(acts, grads, self.out_hw, alpha, self.variant, thresh)

    def generate_coalesced(self, batches, alpha=1.0, class_idxs=None, thresh=None, streams=3, device_batch=32):
        """The batches of a loader merged into device batches of up to ``device_batch`` images (eval-mode BatchNorm makes images
        independent of their batch), ``streams`` of those in flight, the results handed back per loader batch.  At B=8 a batch
        is ~130 launches of 25-200-workgroup grids, 15-25 us each whatever their size: latency-bound; at B=32 the same launches
        carry four times the pixels (the 14 x 14 layers reach >= 256 workgroups with fewer K slices).  A merged batch has
        its own per-tensor amax scales, tile shapes and K-slice counts, so its CAMs equal the batch-by-batch ones to the fp32
        noise of a 50-layer network (~2e-3 of the min-max normalised map's range, as between any two fp32 runs), not bit for
        bit - the masks agree outside that band (tests/test_hip_models.py::test_cam_batches_in_flight_equal_batch_by_batch)."""
        if class_idxs is None:
            class_idxs = [None] * len(batches)
        if device_batch <= 0 or len(batches) <= 1:
            return self.generate_batches(batches, alpha, class_idxs, thresh, streams)
        if self.model.training or any(getattr(m, "training", False) for m in self.model.modules()):
            raise RuntimeError("generate_coalesced(device_batch > 0): the model must be in eval mode - train-mode BatchNorm makes "
                               "an image's CAM depend on the batch it is merged into")
        merged, cls_merged, spans, cur, cur_cls, n = [], [], [], [], [], 0
        for b, c in zip(batches, class_idxs):
            if cur and (n + b.shape[0] > device_batch or b.shape[1:] != cur[0].shape[1:] or (c is None) != (cur_cls[0] is None)):
                merged.append(cur); cls_merged.append(cur_cls); cur, cur_cls, n = [], [], 0
            spans.append((len(merged), n, b.shape[0]))
            cur.append(b); cur_cls.append(c); n += b.shape[0]
        merged.append(cur); cls_merged.append(cur_cls)
        big = [m[0] if len(m) == 1 else torch.cat(m) for m in merged]
        big_cls = [None if c[0] is None else (c[0].view(-1) if len(c) == 1 else torch.cat([t.view(-1).to(b.device) for t in c]))
                   for c, b in zip(cls_merged, big)]
        outs = self.generate_batches(big, alpha, big_cls, thresh, streams)
        res = []
        for j, off, cnt in spans:
            o = outs[j]
            res.append(tuple(t[off:off + cnt] for t in o) if isinstance(o, tuple) else o[off:off + cnt])
        return res

    def generate_batches(self, batches, alpha=1.0, class_idxs=None, thresh=None, streams=3, graphs=None):
        """Several independent batches in flight: batch j runs on stream j % ``streams`` with a generator of its own over
        the SAME model.  At B=8 and 224x224 a batch is ~130 launches of 25-100 workgroups - a fraction of the chip, and
        latency-bound; images are independent (eval-mode BatchNorm), so the batches of a loader can overlap.  Returns
        the list of ``generate_batch`` results, identical to calling it batch by batch.

        ``graphs`` (default: on for the staged path, WSDL_CAM_GRAPH=0 turns it off): each lane captures its batch once
        per (shape, arguments) into a hipGraph and replays it.  Issuing a batch eagerly costs the host ~2.8 ms (Python,
        ctypes and autograd around ~130 launches) - as long as the GPU needs for it - so three eager lanes were bound by
        the one host thread; a replay is one host call, and the lanes really overlap on the device.  Same kernels, same
        order: the results are those of the eager path bit for bit."""
        if class_idxs is None:
            class_idxs = [None] * len(batches)
        if streams <= 1 or len(batches) <= 1 or not batches[0].is_cuda:
            return [self.generate_batch(b, alpha, c, thresh) for b, c in zip(batches, class_idxs)]
        if graphs is None:
            graphs = self.staged and os.environ.get("WSDL_CAM_GRAPH", "1") != "0"
        dev = batches[0].device
        lanes = self._get_lanes(streams, dev)
        cur = torch.cuda.current_stream(dev)
        outs = []
        state = self._state_key() if graphs else None
        for j, (imgs, cls) in enumerate(zip(batches, class_idxs)):
            lane = lanes[j % streams]
            st, gen = lane["stream"], lane["gen"]
            st.wait_stream(cur)                              # the inputs (and the cached weight layouts) are ready
            imgs.record_stream(st)
            if cls is not None:
                cls = cls.to(dev).view(-1)
                cls.record_stream(st)
            with torch.cuda.stream(st):
                out = self._lane_batch(lane, imgs, float(alpha), cls, thresh, state) if graphs else gen.generate_batch(imgs, alpha, cls, thresh)
            for t in (out if isinstance(out, tuple) else (out,)):
                t.record_stream(cur)
            outs.append(out)
        for lane in lanes[:streams]:
            cur.wait_stream(lane["stream"])
        return outs

    def _get_lanes(self, n, dev):
        """The generator's lanes (stream + a plain generator over the same model each), created on demand.  The process
        should not hold more streams than it has hardware queues (4 by default on ROCm): with a fifth stream in use two lanes
        shared a queue and three batches in flight took 0.27 ms/img instead of 0.185 - so the lanes run on the library's
        streams (``ops.lane_stream``: the side stream, the prep stream, then one more) and the one-batch graph of
        ``generate_batch`` is captured on lane 0's stream instead of on a stream of its own."""
        lanes = self.__dict__.setdefault("_lanes", [])
        while len(lanes) < n:
            lanes.append({"stream": ops.lane_stream(dev, len(lanes)), "graph": None, "key": None,
                          "gen": LayerCAMGenerator(self.model, self.target_layer_names, self.variant, self.out_hw, self.staged,
                                                   auto_graph=False, fc_param_grads=self.fc_param_grads)})
        return lanes

    @staticmethod
    def _lane_batch(lane, imgs, alpha, cls, thresh, state=None):
        """One batch on a lane through its hipGraph (current stream = the lane's): eager once (allocator, weight-layout
        caches and lazy library state settle), then capture, then replay with the inputs copied into the captured
        buffers; the outputs are cloned out of the graph's memory before the next replay overwrites them."""
        gen = lane["gen"]
        key = (tuple(imgs.shape), cls is not None, float(alpha), thresh, state)
        warm = lane.setdefault("warm_shapes", set())
        if key[:2] not in warm:                     # a shape's first batch runs eagerly (lazy library state, allocator)
            warm.add(key[:2])
            return gen._generate_batch_eager(imgs, alpha, cls, thresh)
        if lane["graph"] is None or lane["key"] != key:
            dev = imgs.device
            st = torch.cuda.current_stream(dev)
            # capture needs a non-default stream; a lane already is one, the caller's own stream may be the default stream
            cap = st if st != torch.cuda.default_stream(dev) else lane.get("cap_stream", st)
            lane["s_imgs"] = imgs.clone()
            lane["s_cls"] = cls.clone() if cls is not None else None
            st.synchronize()
            ops.reset_amax_pool(dev)                # the first slot request inside the capture allocates + zeroes a pool there
            g = torch.cuda.CUDAGraph()
            try:
                with torch.cuda.graph(g, stream=cap):
                    lane["out"] = gen._generate_batch_eager(lane["s_imgs"], alpha, lane["s_cls"], thresh)
            finally:
                ops.reset_amax_pool(dev)            # eager code must not hand out the graph's slots
            lane["graph"], lane["key"] = g, key
            lane["ws"] = ops._ws_cache.get((dev, cap.cuda_stream))     # the captured launches keep pointing into this buffer
        else:
            lane["s_imgs"].copy_(imgs)
            if cls is not None:
                lane["s_cls"].copy_(cls)
        lane["graph"].replay()
        out = lane["out"]
        return tuple(t.clone() for t in out) if isinstance(out, tuple) else out.clone()

    def generate(self, images, alpha=1.0, class_idx=None):
        """images (3,H,W) -> (1,outH,outW), as the reference (unsqueeze inside)."""
        if torch.is_tensor(alpha) and not torch.is_tensor(class_idx):     # notebook order (img, class_idx, alpha)
            alpha, class_idx = (1.0 if class_idx is None else class_idx), alpha
        x = images.unsqueeze(0) if images.dim() == 3 else images
        return self.generate_batch(x, float(alpha), class_idx)

    __call__ = generate

    @staticmethod
    def bg_from_cams(all_cams, alpha=2.0, out_hw=(224, 224)):
        """(n, H, W) maps -> (m_bg, max_obj): maximum over the leading axis, ``1 - clamp(1 - max, 0) ** alpha``, both
        bilinearly resized to ``out_hw`` (reference AlternatingDirectionCutLoss.py:304-318)."""
        max_obj = all_cams.max(dim=0).values
        m_bg = 1.0 - ((1.0 - max_obj).clamp(min=0.0) ** alpha)
        both = ops.bilinear_resize(torch.stack([m_bg, max_obj]).unsqueeze(0).contiguous(), out_hw)[0]
        return both[0], both[1]

    def generate_bg_cam(self, image_tensor, valid_class_indices, alpha=2.0):
        """``LayerCAMGenerator.generate_bg_cam`` of the notebook class (reference
        TraditionalModel/AlternatingDirectionCutLoss.py:296-318; "mimics CAMGenerator's bg+fg map output"):
        ``generate(image, valid_class_indices)`` - the notebook argument order, alpha 1.0 - then the maximum over the map's
        leading axis, the background map ``1 - (1 - max)**alpha`` and the resize of both to 224 x 224.  As in the
        reference, the class-score gather accepts ONE valid class per image (more raise there too)."""
        idx = torch.as_tensor(valid_class_indices, device=image_tensor.device)
        with torch.no_grad():
            all_cams = self.generate(image_tensor, 1.0, class_idx=idx)
            return self.bg_from_cams(all_cams, alpha, (224, 224))


@torch.no_grad()
def evaluate_layercam_on_test_set(layercam_gen, test_loader, alpha=1.0, cam_thresh=0.3, *, device="cuda", max_images=11,
                                  log=print):
    """Reference ``evaluate_layercam_on_test_set(layercam_gen, test_loader, alpha=1.0, cam_thresh=0.3)``
    (LayerCAM.py:84-130): IoU / pixel accuracy of the thresholded LayerCAM foreground mask against the ground-truth
    trimap (foreground = class 1), one image - the first of each batch - per batch, over the first 11 batches (the
    reference breaks after ``i >= 10``, :119-120), nearest resize of the prediction when sizes differ.  Returns
    ``{"layercam_fg_iou", "layercam_fg_acc"}``.  The threshold is fused into the CAM epilogue kernel; the metric is
    the host-side ``compute_iou_and_acc`` of the reference."""
    from .ExtraUtilities import compute_iou_and_acc
    ious, accs = [], []
    for i, (img, (label, true_mask)) in enumerate(test_loader):
        x = img[0].to(device)
        tm = (true_mask[0].to(device) == 1).long()
        tm = tm.reshape(tm.shape[-2:])
        lab = label[0]
        cls = torch.as_tensor([int(lab.item() if torch.is_tensor(lab) else lab)], device=device)
        _cam, mask = layercam_gen.generate_batch(x.unsqueeze(0), float(alpha), cls, thresh=cam_thresh)
        pred = mask[0].long()
        if pred.shape != tm.shape:
            # F.interpolate(mode='nearest'): source index = floor(dst * in / out)
            ih = torch.arange(tm.shape[0], device=device) * pred.shape[0] // tm.shape[0]
            iw = torch.arange(tm.shape[1], device=device) * pred.shape[1] // tm.shape[1]
            pred = pred[ih][:, iw]
        iou, acc = compute_iou_and_acc(pred, tm)
        ious.append(iou)
        accs.append(acc)
        if max_images is not None and i >= max_images - 1:
            break
    out = {"layercam_fg_iou": sum(ious) / len(ious), "layercam_fg_acc": sum(accs) / len(accs)}
    if log:
        log("\n Evaluation of CAMs on test set:")
        log(f" - LayerCam FG: Avg IoU: {out['layercam_fg_iou']:.4f} | Acc: {out['layercam_fg_acc']:.4f}")
    return out


class CAMGenerator:
    """Classic fc-weight CAM (reference TraditionalModel/AlternatingDirectionCutLoss.py:320-403).

    ``generate_all_cams(image (3,H,W)) -> (num_classes, h, w)``: the per-class ``einsum("c,chw->hw")`` loop of the
    reference is ONE 1x1 convolution of layer4's features with ``fc.weight`` (37 x 2048 . 2048 x hw on the MFMA
    kernel), followed by the ReLU + per-class min-max kernel.  ``generate_bg_cam`` mirrors the reference: masked
    max over the valid classes, ``1 - (1 - max)**alpha``, both maps bilinearly resized to 224 x 224."""

    def __init__(self, model):
        self.model = model.eval()

    @torch.no_grad()
    def generate_all_cams_batch(self, images):
        _logits, feats = self.model(images)
        f = feats[-1]
        w = self.model.fc.weight
        raw = ops.conv_bias_act(f, w.reshape(w.shape[0], w.shape[1], 1, 1))      # (B, classes, h, w)
        return ops.plane_relu_minmax(raw)

    def generate_all_cams(self, image_tensor):
        return self.generate_all_cams_batch(image_tensor.unsqueeze(0))[0]

    @torch.no_grad()
    def generate_bg_cam(self, image_tensor, valid_class_indices, alpha=1.0, out_hw=(224, 224)):
        cams = self.generate_all_cams(image_tensor)
        idx = torch.as_tensor(list(valid_class_indices), device=cams.device, dtype=torch.long)
        max_obj = cams.index_select(0, idx).amax(dim=0).clamp(min=0.0)          # masked-out classes contribute 0
        m_bg = 1.0 - ((1.0 - max_obj).clamp(min=0.0) ** alpha)
        both = ops.bilinear_resize(torch.stack([m_bg, max_obj]).unsqueeze(0).contiguous(), out_hw)[0]
        return both[0], both[1]
