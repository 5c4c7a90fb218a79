"""SegmentationModel (DeepLabV3-ResNet50) on the HIP path.

The reference has no class of this name: it builds
``torchvision.models.segmentation.deeplabv3_resnet50(pretrained=True)`` and swaps
``classifier[4] = nn.Conv2d(256, 2, 1)`` (TraditionalModel/SegmentationModel.py:85-88,
AlternatingDirectionCutLoss.py:784-787), always calling ``model(images)['out']``.  ``SegmentationModel``
keeps that call surface - ``forward(x)`` returns an OrderedDict with 'out' (and 'aux' while the aux
head exists) - and torchvision's module / state_dict names (backbone.*, classifier.0.convs.*,
classifier.0.project.*, classifier.{1,2,4}.*, aux_classifier.{0,1,4}.*).

Architecture (published torchvision definition): ResNet-50 backbone with
``replace_stride_with_dilation=[False, True, True]`` (output stride 8), DeepLabHead = ASPP(2048,
rates 12/24/36, + image pooling) -> 1x1 project + Dropout(0.5) -> 3x3 -> 1x1, FCNHead aux classifier
on layer3 (computed every forward as in the reference, unused by its losses), bilinear up-sampling to
the input size.  Convolutions, BatchNorm, pooling, dropout, resampling and the loss are HIP kernels.

``train_step`` is one iteration of the reference's training loop (SegmentationModel.py:96-113 /
AlternatingDirectionCutLoss.py:693-703): clamp masks to {0,1}, forward, CrossEntropy, backward, Adam.
"""
from collections import OrderedDict

import os

import torch
import torch.nn as nn

from .. import nn as wnn
from .. import ops
from .. import plan
from ..optim import FlatAdam
from .ExtraUtilities import compute_iou_and_acc


def _cbr(cin, cout, k, dilation=1):
    return [wnn.Conv2d(cin, cout, k, padding=0 if k == 1 else dilation, dilation=dilation), wnn.BatchNorm2d(cout),
            wnn.ReLU()]


class _ASPPPooling(wnn.FusedSequential):
    def __init__(self, cin, cout):
        super().__init__(wnn.GlobalAvgPool(), *_cbr(cin, cout, 1))

    def forward(self, x):
        size = x.shape[-2:]
        return ops.bilinear_resize(super().forward(x), size)


_ASPP_IN_PLACE = os.environ.get("WSDL_ASPP_IN_PLACE", "1") != "0"     # A/B: 0 = plane copies of all five branches


class ASPP(nn.Module):
    def __init__(self, cin=2048, rates=(12, 24, 36), cout=256):
        super().__init__()
        branches = [wnn.FusedSequential(*_cbr(cin, cout, 1))]
        branches += [wnn.FusedSequential(*_cbr(cin, cout, 3, r)) for r in rates]
        branches.append(_ASPPPooling(cin, cout))
        self.convs = nn.ModuleList(branches)
        self.project = wnn.FusedSequential(*_cbr(len(branches) * cout, cout, 1), wnn.Dropout(0.5))

    def forward(self, x):
        # x feeds five branches: chain it through the four conv nodes (ops.conv_bn_act passthrough) so that the five
        # input gradients are summed inside the dgrad epilogues instead of by four 134 MB autograd adds
        branches = list(self.convs)
        if x.is_cuda and all(b[1].training for b in branches[:-1]) and _ASPP_IN_PLACE:
            # train mode: the BatchNorm kernels of the four convolution branches write straight into their channel slices of
            # the concatenation (and publish into ONE amax slot); only the pooling branch is copied in
            B, _, H, W = x.shape
            cs = [b[0].out_channels for b in branches[:-1]]
            cat = torch.empty(B, self.project[0].in_channels, H, W, device=x.device, dtype=torch.float32)
            slot = ops.amax_slot(x.device) if ops.CONV_ARITH[0] == 1 else None
            outs, off = [], 0
            convs = [b[0] for b in branches[:-1]]
            bns = [b[1] for b in branches[:-1]]
            if (ops.ASPP_MULTI[0] and torch.is_grad_enabled() and x.requires_grad and len(set(cs)) == 1
                    and all(m.stride == 1 and m.bias is None and m.weight.requires_grad and m.padding == m.dilation * (m.kernel_size - 1) // 2
                            and getattr(m.weight, "_wsdl_grad_sink", None) is not None for m in convs)
                    and all(bn.weight.requires_grad and getattr(bn.weight, "_wsdl_grad_sink", None) is not None
                            and bn.momentum == bns[0].momentum and bn.eps == bns[0].eps for bn in bns)
                    and ops.dgrad_multi_ok(len(convs), tuple(x.shape), cs[0])):
                # the four convolution branches as ONE node: their input gradients are one launch (ops._ConvBNBranches)
                holders = []
                for c in cs:
                    holders.append([cat[:, off:off + c], slot])
                    off += c
                for bn in bns:
                    bn._pending_steps += 1
                res = ops.conv_bn_branches(x, list(zip(convs, bns, holders)), bns[0].momentum, bns[0].eps)
                outs, x = list(res[:-1]), res[-1]
                outs.append(branches[-1](x))
                return self.project(ops.concat_into(cat, slot, outs))
            for b, c in zip(branches[:-1], cs):
                y, x = wnn.conv_bn(x, b[0], b[1], True, passthrough=True, out_holder=[cat[:, off:off + c], slot])
                outs.append(y)
                off += c
            outs.append(branches[-1](x))
            return self.project(ops.concat_into(cat, slot, outs))
        outs = []
        for b in branches[:-1]:
            y, x = wnn.conv_bn(x, b[0], b[1], True, passthrough=True)
            outs.append(y)
        outs.append(branches[-1](x))
        return self.project(ops.concat_channels(outs))


class _Backbone(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv1, self.bn1, (self.layer1, self.layer2, self.layer3, self.layer4) = \
            wnn.make_resnet50_stages((False, True, True))
        self.relu, self.maxpool = wnn.ReLU(), wnn.MaxPool3x3s2()

    def forward(self, x):
        x = self.maxpool(wnn.conv_bn(x, self.conv1, self.bn1, True))
        x = self.layer2(self.layer1(x))
        f3 = self.layer3(x)
        return OrderedDict(out=self.layer4(f3), aux=f3)


_AUX_ON_SIDE_STREAM = os.environ.get("WSDL_AUX_SIDE", "1") != "0"


class _Outputs(OrderedDict):
    """The result dict of ``forward``.  In eval mode the aux head - which no caller of the reference ever reads
    ("computed every forward, unused by every loss") and which has no side effect there (eval-mode
    BatchNorm, no dropout) - is computed when 'aux' is first looked at instead of on every forward: refine_pseudo_mask
    and evaluate_model only take ['out'].  In train mode it is computed eagerly, as torchvision does (its BatchNorm
    running statistics move)."""

    def __init__(self):
        super().__init__()
        self._lazy = {}

    def _resolve(self, key):
        fn = self._lazy.pop(key, None)
        if fn is not None:
            super().__setitem__(key, fn())

    def __getitem__(self, key):
        self._resolve(key)
        return super().__getitem__(key)

    def __contains__(self, key):
        return key in self._lazy or super().__contains__(key)

    def _all(self):
        for k in list(self._lazy):
            self._resolve(k)

    def keys(self):
        self._all()
        return super().keys()

    def values(self):
        self._all()
        return super().values()

    def items(self):
        self._all()
        return super().items()

    def __iter__(self):
        self._all()
        return super().__iter__()

    def __len__(self):
        return super().__len__() + len(self._lazy)

    # every read access of the dict API resolves the lazy entries first: ``out.get('aux')`` must not answer None for a
    # key ``'aux' in out`` reports (torchvision returns a plain OrderedDict)
    def get(self, key, default=None):
        return self[key] if key in self else default

    def pop(self, key, *default):
        self._resolve(key)
        return super().pop(key, *default)

    def popitem(self, last=True):
        self._all()
        return super().popitem(last)

    def setdefault(self, key, default=None):
        if key in self:
            return self[key]
        super().__setitem__(key, default)
        return default

    def __setitem__(self, key, value):
        if getattr(self, "_lazy", None):
            self._lazy.pop(key, None)
        super().__setitem__(key, value)

    def __delitem__(self, key):
        if self._lazy.pop(key, None) is None:
            super().__delitem__(key)

    def copy(self):
        self._all()
        return OrderedDict(super().items())

    def __eq__(self, other):
        self._all()
        return super().__eq__(other)

    def __repr__(self):
        self._all()
        return super().__repr__()


class SegmentationModel(nn.Module):
    def __init__(self, num_classes=2, aux_loss=True, aux_classes=21):
        super().__init__()
        self.backbone = _Backbone()
        self.classifier = wnn.FusedSequential(ASPP(), *_cbr(256, 256, 3, 1), wnn.Conv2d(256, num_classes, 1, bias=True))
        self.aux_classifier = None
        if aux_loss:
            self.aux_classifier = wnn.FusedSequential(*_cbr(1024, 256, 3, 1), wnn.Dropout(0.1),
                                                      wnn.Conv2d(256, aux_classes, 1, bias=True))
        for m in self.modules():
            if isinstance(m, wnn.Conv2d) and m.bias is not None:
                m.reset_parameters()

    # The train-mode aux head runs on the side stream (forward below) and writes its BatchNorm running statistics there.
    # A training step joins that stream before Adam; a train-mode forward WITHOUT a step (BatchNorm recalibration, a
    # loss-only validation pass) does not - so every place that reads those buffers on another stream joins first.
    def _join_aux(self):
        ev = self.__dict__.get("_aux_event")
        if ev is not None:
            self.__dict__["_aux_event"] = None
            if not torch.cuda.is_current_stream_capturing():
                ev.wait()      # the event sits right behind the aux head: not the whole side stream

    def train(self, mode=True):
        self._join_aux()
        return super().train(mode)

    def state_dict(self, *args, **kwargs):
        self._join_aux()
        return super().state_dict(*args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        self._join_aux()
        return super().load_state_dict(*args, **kwargs)

    def forward(self, x):
        size = x.shape[-2:]
        self._join_aux()              # a previous forward's aux head may still be writing the statistics read / written now
        feats = self.backbone(x)
        res = _Outputs()
        if self.aux_classifier is not None and self.training and x.is_cuda and _AUX_ON_SIDE_STREAM and \
                not torch.cuda.is_current_stream_capturing():
            # Train mode computes the aux head on every forward, as torchvision does (its BatchNorm statistics move), and
            # no loss of the reference reads it: it runs on the side stream - idle during forward - beside the main head
            # instead of after it; whoever does look at ['aux'] joins first.
            f3 = feats["aux"]
            side = ops.side_stream(x.device)
            ops.stream_wait(side, ops.raw_stream(x.device))
            ops.cross_stream_use(f3, side)
            with torch.cuda.stream(side):
                aux_out = ops.bilinear_resize(self.aux_classifier(f3), size)
            ev = self.__dict__.get("_aux_event_obj")        # one event, re-recorded by every forward (part of a launch plan)
            if ev is None:
                ev = self.__dict__["_aux_event_obj"] = ops.Event()
            ev.record(side)
            self.__dict__["_aux_event"] = ev

            def joined():
                ops.stream_wait(ops.raw_stream(x.device), side)
                return aux_out
            res["out"] = ops.bilinear_resize(self.classifier(feats["out"]), size)
            res._lazy["aux"] = joined
            return res
        res["out"] = ops.bilinear_resize(self.classifier(feats["out"]), size)
        if self.aux_classifier is not None:
            f3 = feats["aux"]
            if self.training:
                res["aux"] = ops.bilinear_resize(self.aux_classifier(f3), size)
            else:
                grad = torch.is_grad_enabled()

                def aux():
                    with torch.set_grad_enabled(grad):
                        return ops.bilinear_resize(self.aux_classifier(f3), size)
                res._lazy["aux"] = aux
        return res


def build_segmentation_model(num_classes=2, aux_loss=True):
    """The reference's construction: 21-class DeepLabV3 with aux head, classifier[4] swapped."""
    return SegmentationModel(num_classes=num_classes, aux_loss=aux_loss, aux_classes=21)


def resolve_criterion(criterion):
    """What a reference-style ``criterion`` object means on the HIP path.  ``nn.CrossEntropyLoss()`` (the only criterion
    the reference constructs: SegmentationModel.py:90, AlternatingDirectionCutLoss.py:789) maps onto the fused
    softmax-cross-entropy kernel with its ``ignore_index``; options the kernel does not implement raise instead of being
    dropped silently.  Any other callable is applied to ``(outputs, masks)`` as it is."""
    if criterion is None:
        return lambda o, m: ops.cross_entropy(o, m.long())
    if isinstance(criterion, nn.CrossEntropyLoss):
        if criterion.weight is not None or criterion.reduction != "mean" or getattr(criterion, "label_smoothing", 0.0) != 0.0:
            raise ValueError("train_step: nn.CrossEntropyLoss with class weights, a reduction other than 'mean' or label "
                             "smoothing is not implemented by wsdl_softmax_ce_fwd_bwd (the reference uses the defaults)")
        ign = int(criterion.ignore_index)
        return lambda o, m: ops.cross_entropy(o, m.long(), ign)
    if callable(criterion):
        return criterion
    raise TypeError(f"criterion: expected nn.CrossEntropyLoss, a callable or None, got {type(criterion).__name__}")


_ONES = {}


def _one(device):
    """The gradient of the loss with respect to itself, allocated once (``loss.backward()`` fills a new tensor with a kernel
    of the tensor library on every call - which a launch plan would not see)."""
    t = _ONES.get(device)
    if t is None:
        t = _ONES[device] = torch.ones((), device=device, dtype=torch.float32)
    return t


def train_step(model, optimizer, images, masks, extra_loss=None, loss_fn="cross_entropy", criterion=None):
    """One training iteration; returns the (device) loss tensor, no host synchronisation.  ``loss_fn``: 'cross_entropy' or
    'lovasz_softmax' (reference SegmentationModel.py:65,103-107); ``criterion``: a reference-style loss object instead
    (``resolve_criterion``).

    On the device, in train mode and outside data parallelism the iteration is issued as ONE host call from its third
    occurrence on (``plan.PlannedTrainStep``: the launches of an eager iteration recorded behind the C ABI, verified to
    reproduce it bit for bit, then replayed); WSDL_PLAN_STEP=0 keeps every iteration eager."""
    if isinstance(optimizer, FlatAdam) and images.is_cuda and plan.PLAN_STEP[0]:
        tag = (plan.loss_tag(extra_loss), loss_fn, plan.loss_tag(criterion))
        st = plan.planned_step_for(model, optimizer,
                                   lambda i, m: _train_step_eager(model, optimizer, i, m, extra_loss, loss_fn, criterion), tag)
        # host scalars inside the loss objects (weights, window sizes, sigmas) are kernel arguments a plan freezes: part of its key
        st.loss_scalars = (plan.host_scalars(extra_loss), plan.host_scalars(criterion)) if (extra_loss is not None or criterion is not None) else None
        return st(images, masks)
    return _train_step_eager(model, optimizer, images, masks, extra_loss, loss_fn, criterion)


def _train_step_eager(model, optimizer, images, masks, extra_loss=None, loss_fn="cross_entropy", criterion=None):
    masks = ops.clamp_max_labels(masks, 1)
    with ops.prof_range("train_step/forward"):
        outputs = model(images)["out"]
    with ops.prof_range("train_step/loss"):
        outputs_x = outputs
        if extra_loss is not None and outputs.is_cuda:
            # two consumers of the logits: their gradients are summed by the library (ops.fanout), not by the autograd
            # engine's own add kernel - which a launch plan would not see
            outputs, outputs_x = ops.fanout(outputs, 2)
        if criterion is not None:
            loss = resolve_criterion(criterion)(outputs, masks)
        elif loss_fn == "lovasz_softmax":
            loss = ops.lovasz_softmax(ops.softmax_channels(outputs), masks.long(), classes="present", per_image=False, ignore=None)
        elif loss_fn == "cross_entropy":
            loss = ops.cross_entropy(outputs, masks.long())
        else:
            raise ValueError(f"loss_fn {loss_fn!r}: 'cross_entropy' or 'lovasz_softmax'")
        if extra_loss is not None:
            extra = extra_loss(outputs_x, images)
            if torch.is_tensor(extra) and extra.is_cuda and extra.dim() == 0 and loss.dim() == 0 and extra.dtype == loss.dtype:
                loss = ops.add_scalars(loss, extra)      # a launch of the library (visible to a launch plan)
            else:
                loss = loss + extra
    with ops.prof_range("train_step/backward"):
        optimizer.zero_grad()
        loss.backward(_one(loss.device) if (loss.dim() == 0 and loss.dtype == torch.float32 and loss.is_cuda) else None)
    with ops.prof_range("train_step/optimizer"):
        optimizer.step()
    return loss.detach()


def make_optimizer(model, lr=1e-4, early_step=None):
    """torch.optim.Adam(model.parameters(), lr) semantics on one flat buffer.  Single process: one Adam launch per step,
    then the re-layout of every convolution weight on the side stream.  Under ``dp.GradBucketReducer`` the buffer is
    stepped in segments (optim.FlatAdam.enable_early_step): each segment's Adam launch and the re-layout of its
    convolution weights follow its gradient all-reduce.  ``early_step=True`` (or WSDL_EARLY_STEP=1) uses the segments
    in a single process too."""
    params = [p for p in model.parameters() if p.requires_grad]
    opt = FlatAdam(params, lr=lr)
    convs = [m for m in model.modules() if isinstance(m, wnn.Conv2d) and m.weight.requires_grad and m.weight.is_cuda]
    if convs:
        index = {id(p): i for i, p in enumerate(params)}
        by_param = {index[id(m.weight)]: m for m in convs}

        def relayout(k, param_indices, use_events, adam_event=None):
            if k == 0:
                adam_event = None        # the first layers' layouts gate the next forward: not behind the other segments' queue
            seg_convs = [by_param[i] for i in param_indices if i in by_param]
            # runs before step() bumps the parameter epoch; in eager mode into the spare buffers (this step's remaining
            # input-gradient kernels still read the current ones), inside a captured graph in place
            return ops.prefetch_weight_layouts(seg_convs, use_events=use_events, epoch_ahead=1, pingpong=use_events,
                                               after=adam_event)

        opt.post_step_hook = lambda: ops.prefetch_weight_layouts(convs)     # after the single-launch step
        if early_step is None:
            early_step = {"0": False, "1": True, "2": "tail"}[os.environ.get("WSDL_EARLY_STEP", "0")]
        if early_step == "tail":
            # two segments: [the stem | everything else].  "Everything else" is stepped while the stem's backward (max-pool,
            # BatchNorm over the 67 MB map, the 7x7 weight gradient: 0.3 ms of small launches) still runs
            stem = params[0].numel() + sum(p.numel() for p in params[1:3])
            opt.enable_early_step(relayout, first=(stem + 63) // 64 * 64, growth=1 << 20, cap=1 << 40)
        else:
            opt.enable_early_step(relayout)
        opt.early_step = bool(early_step)
        if opt.early_step:
            opt.register_announce_hooks()
    return opt


@torch.no_grad()
def evaluate_model(model, loader, device="cuda", binarize="notebook"):
    """argmax -> nearest resize to the ground truth's size -> IoU / pixel accuracy, averaged over the loader's images
    (one image - the first of each batch - per batch, as the reference).  Ground truth from the Oxford-IIIT Pet trimap:
      binarize="notebook": ``tm[tm == 2] = 1; tm = 1 - tm``      (AlternatingDirectionCutLoss.py:639-682, the script
                           that actually runs - SURVEY.md section 0);
      binarize="modular" : ``tm = (tm == 1)``                    (SegmentationModel.py:126-159)."""
    model.eval()
    ious, accs = [], []
    for img, (_label, true_mask) in loader:
        x = img[0].to(device).unsqueeze(0)
        tm = true_mask[0].to(device).clone()
        if binarize == "notebook":
            tm[tm == 2] = 1
            tm = 1 - tm
        elif binarize == "modular":
            tm = (tm == 1).long()
        else:
            raise ValueError(binarize)
        out = model(x)["out"]
        pred = out.squeeze(0).argmax(dim=0)
        if pred.shape != tm.shape:
            # F.interpolate(mode='nearest'): source index = floor(dst * in / out)
            idx_h = (torch.arange(tm.shape[-2], device=device) * pred.shape[0] // tm.shape[-2])
            idx_w = (torch.arange(tm.shape[-1], device=device) * pred.shape[1] // tm.shape[-1])
            pred = pred[idx_h][:, idx_w]
        iou, acc = compute_iou_and_acc(pred, tm)
        ious.append(iou)
        accs.append(acc)
    return sum(ious) / len(ious), sum(accs) / len(accs)


def train_segmentation_model(loss_fn, run_id, lr=1e-4, num_epochs=10, batch_size=4, val_split=0.2, *, out_root="/content",
                             device="cuda", val_loader=None, num_workers=0, seed=None, log=print, model=None):
    """Reference ``train_segmentation_model(loss_fn, run_id, lr=1e-4, num_epochs=10, batch_size=4, val_split=0.2)``
    (TraditionalModel/SegmentationModel.py:59-122), same positional signature, returns ``(model, final_loss)``.

    Trains DeepLabV3-ResNet50 (``build_segmentation_model``) with Adam(lr) on the pseudo masks of run ``run_id`` -
    ``{out_root}/images_{run_id}`` / ``{out_root}/pseudo_masks_{run_id}``, what ``generate_pseudo_masks`` wrote - with
    ``loss_fn`` 'cross_entropy' or 'lovasz_softmax'; batches of one image are skipped (:97-98), masks clamped to {0,1} (:100).

    Where the reference's text cannot run, the working notebook decides: the reference builds the
    pseudo-mask dataset (:73-77) and then trains on ``load_split_data()`` (:80-83), whose items are not (image, mask)
    pairs; AlternatingDirectionCutLoss.py:775-781 trains on ``PseudoSegmentationDataset`` - so does this.  ``val_split`` is
    accepted and, as in the reference (which never reads it), unused; the per-epoch validation of :118-119 runs when a
    ``val_loader`` of ``(img, (label, trimap))`` items is given (the reference takes it from the Oxford-IIIT Pet download,
    which needs the network).  Keyword-only extras: the directories' root (the reference hard-codes /content), the
    device, an existing model to continue from."""
    from torch.utils.data import DataLoader
    from .SegmentationDataset import PseudoSegmentationDataset
    if loss_fn not in ("cross_entropy", "lovasz_softmax"):
        raise ValueError(f"loss_fn {loss_fn!r}: 'cross_entropy' or 'lovasz_softmax'")
    import os as _os
    image_dir = _os.path.join(out_root, f"images_{run_id}")
    mask_dir = _os.path.join(out_root, f"pseudo_masks_{run_id}")
    full_dataset = PseudoSegmentationDataset(img_dir=image_dir, mask_dir=mask_dir, transform=True)
    gen = torch.Generator().manual_seed(seed) if seed is not None else None
    train_loader = DataLoader(full_dataset, batch_size=batch_size, shuffle=True, num_workers=num_workers, generator=gen)
    if model is None:
        model = build_segmentation_model(num_classes=2)
    model = model.to(device)
    optimizer = make_optimizer(model, lr=lr)
    final_loss = 0.0
    for epoch in range(num_epochs):
        model.train()
        total = torch.zeros((), device=device)
        for images, masks in train_loader:
            if images.size(0) == 1:
                continue
            total += train_step(model, optimizer, images.to(device), masks.to(device), loss_fn=loss_fn)
        final_loss = total.item()                       # one host read per epoch (the reference: one per step)
        if log:
            log(f"[Run {run_id}] Epoch {epoch + 1}/{num_epochs}, Loss: {final_loss:.4f}")
        if val_loader is not None:
            avg_iou, avg_acc = evaluate_model(model, val_loader, device=device, binarize="modular")
            if log:
                log(f"[Run {run_id}] Validation IoU: {avg_iou:.4f}, Accuracy: {avg_acc:.4f}")
    return model, final_loss
