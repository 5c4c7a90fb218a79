#!/usr/bin/env python3
"""Generate the golden fixtures in this directory from the reference's own function bodies.

Runs ONLY in the build container (needs /root/reference, which never travels to the GPU box).
The reference modules cannot be imported (missing torchvision / skimage, wrong module names,
a notebook export that trains at import time - SURVEY.md section 0.2 / 8c), so individual
ClassDef / FunctionDef nodes are selected by name with ``ast`` and executed in a namespace that
provides torch / nn / F.  Nothing of the reference's text is written to the fixtures: they hold
seeded inputs and the outputs the reference bodies produced for them.

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz

``keep_largest`` needs skimage, which only the stale /opt/conda/bin/python3.9 has; that part is
run in a subprocess of that interpreter.

Known reference defect handled here: ``ConstrainToBoundaryLossSingle.
compute_affinities_single`` is declared without ``self`` but invoked through ``self``; the
fixture is produced with the method re-bound as a staticmethod (the intended semantics), the
``forward`` body itself runs unchanged.
"""
import ast
import gc
import json
import os
import subprocess
import sys
import textwrap

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

REF = "/root/reference/TraditionalModel"
HERE = os.path.dirname(os.path.abspath(__file__))


def lift(path, names, extra_ns=None):
    """exec the named top-level ClassDef/FunctionDef nodes of ``path``; return the namespace."""
    tree = ast.parse(open(path).read())
    picked = [n for n in tree.body
              if isinstance(n, (ast.ClassDef, ast.FunctionDef)) and n.name in names]
    # a later definition of the same name shadows an earlier one, as it would at import
    ns = {"torch": torch, "nn": nn, "F": F, "gc": gc, "np": np}
    ns.update(extra_ns or {})
    exec(compile(ast.Module(body=picked, type_ignores=[]), path, "exec"), ns)
    return ns


def smooth_image(B, H, W, seed):
    """Piece-wise smooth RGB in [0,1]: iid pixels make the affinity underflow."""
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.linspace(0, 1, H), torch.linspace(0, 1, W), indexing="ij")
    img = torch.zeros(B, 3, H, W)
    for b in range(B):
        for c in range(3):
            for _ in range(4):
                fy, fx, ph = (torch.rand(3, generator=g) * torch.tensor([3.0, 3.0, 6.28])).tolist()
                img[b, c] += 0.25 * torch.sin(6.28 * (fy * yy + fx * xx) + ph)
    img = img * 0.5 + 0.5 + 0.01 * torch.randn(B, 3, H, W, generator=g)
    return img.clamp(0, 1)


def gen_losses():
    cut = lift(f"{REF}/AlternatingDirectionCutLoss.py", {"LocalNormalizedCutLoss", "compute_affinities"})
    bnd = lift(f"{REF}/AlternatingDirectionBoundaryLoss.py", {"ConstrainToBoundaryLossSingle"})
    Bnd = bnd["ConstrainToBoundaryLossSingle"]
    Bnd.compute_affinities_single = staticmethod(Bnd.__dict__["compute_affinities_single"])  # D1
    out = {}
    cases = [  # (B,C,H,W, sigma_color, window)
        (2, 2, 32, 32, 0.1, 5), (1, 2, 17, 23, 0.05, 5), (2, 3, 16, 16, 0.1, 3), (2, 2, 24, 40, 0.05, 3),
    ]
    meta = []
    for i, (B, C, H, W, sc, w) in enumerate(cases):
        g = torch.Generator().manual_seed(100 + i)
        img = smooth_image(B, H, W, 200 + i)
        preds = (2.0 * torch.randn(B, C, H, W, generator=g)).requires_grad_()
        loss = cut["LocalNormalizedCutLoss"](sigma_color=sc, window_size=w)(preds, img)
        loss.backward()
        out[f"ncut{i}_preds"], out[f"ncut{i}_image"] = preds.detach().numpy(), img.numpy()
        out[f"ncut{i}_loss"], out[f"ncut{i}_grad"] = loss.detach().numpy(), preds.grad.numpy()
        meta.append(dict(kind="ncut", idx=i, sigma_color=sc, window=w))
    # 3-D path of the same forward (unsqueeze) - AlternatingDirectionCutLoss.py:72-74
    g = torch.Generator().manual_seed(110)
    img = smooth_image(1, 20, 28, 210)[0]
    preds = torch.randn(2, 20, 28, generator=g).requires_grad_()
    loss = cut["LocalNormalizedCutLoss"](sigma_color=0.1, window_size=5)(preds, img)
    loss.backward()
    out["ncut3d_preds"], out["ncut3d_image"] = preds.detach().numpy(), img.numpy()
    out["ncut3d_loss"], out["ncut3d_grad"] = loss.detach().numpy(), preds.grad.numpy()

    img = smooth_image(2, 16, 16, 220)
    aff = cut["compute_affinities"](img, sigma_color=0.1, sigma_space=5, window_size=5)
    out["aff_image"], out["aff_maps"] = img.numpy(), torch.stack(aff).numpy()      # (24,2,1,16,16)

    for i, (C, H, W, sc, ss, w) in enumerate([(2, 24, 24, 0.1, 5.0, 5), (3, 15, 19, 0.1, 10.0, 5),
                                              (2, 16, 16, 0.05, 5.0, 3)]):
        g = torch.Generator().manual_seed(120 + i)
        img = smooth_image(1, H, W, 230 + i)[0]
        probs = F.softmax(2.0 * torch.randn(C, H, W, generator=g), dim=0).requires_grad_()
        loss = Bnd(sigma_color=sc, sigma_space=ss, window_size=w)(probs, img)
        loss.backward()
        out[f"bnd{i}_preds"], out[f"bnd{i}_image"] = probs.detach().numpy(), img.numpy()
        out[f"bnd{i}_loss"], out[f"bnd{i}_grad"] = loss.detach().numpy(), probs.grad.numpy()
        meta.append(dict(kind="boundary", idx=i, sigma_color=sc, sigma_space=ss, window=w,
                         note="D1: compute_affinities_single bound as staticmethod"))
    out["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(f"{HERE}/losses.npz", **out)
    print("losses.npz", len(out))


class ToyCAMNet(nn.Module):
    """Small stand-in exposing layer3 / layer4 like FrozenResNetCAM (fixture model, our own)."""

    def __init__(self, c3=48, c4=96, nc=7):
        super().__init__()
        self.stem = nn.Sequential(nn.Conv2d(3, 16, 3, 2, 1), nn.ReLU(), nn.Conv2d(16, 32, 3, 2, 1), nn.ReLU())
        self.layer3 = nn.Sequential(nn.Conv2d(32, c3, 3, 2, 1), nn.ReLU())
        self.layer4 = nn.Sequential(nn.Conv2d(c3, c4, 3, 1, 2, dilation=2), nn.ReLU())
        self.fc = nn.Linear(c4, nc)

    def forward(self, x):
        f3 = self.layer3(self.stem(x))
        f4 = self.layer4(f3)
        return self.fc(f4.mean(dim=(2, 3))), [f3, f4]


def gen_layercam():
    mod = lift(f"{REF}/LayerCAM.py", {"LayerCAMGenerator"})["LayerCAMGenerator"]
    nb = lift(f"{REF}/AlternatingDirectionCutLoss.py", {"LayerCAMGenerator"})["LayerCAMGenerator"]
    torch.manual_seed(7)
    net = ToyCAMNet()
    out = {"state/" + k: v.numpy() for k, v in net.state_dict().items()}
    g = torch.Generator().manual_seed(8)
    imgs = torch.rand(2, 3, 112, 112, generator=g)          # stride 8 -> 14x14 maps
    out["images"] = imgs.numpy()
    out["class_idx"] = np.array([1, 4])
    for vname, cls in (("modular", mod), ("notebook", nb)):
        gen = cls(net, ["layer3", "layer4"])
        for i in range(2):
            for alpha in ((0.5, 1.0, 2.0) if vname == "modular" else (0.5, 2.0)):
                ci = torch.tensor([int(out["class_idx"][i])])
                if vname == "modular":
                    cam = gen.generate(imgs[i], alpha, class_idx=ci)
                else:
                    cam = gen.generate(imgs[i], class_idx=ci, alpha=alpha)
                out[f"{vname}_cam_{i}_a{alpha}"] = cam.numpy()
                if alpha == 1.0 and vname == "modular":
                    for n in ("layer3", "layer4"):
                        out[f"act_{n}_{i}"] = gen.activations[n].detach().numpy()
                        out[f"grad_{n}_{i}"] = gen.gradients[n].detach().numpy()
        # default class (argmax) path
        cam = gen.generate(imgs[0], 1.0) if vname == "modular" else gen.generate(imgs[0])
        out[f"{vname}_cam_argmax"] = cam.numpy()
    np.savez_compressed(f"{HERE}/layercam.npz", **out)
    print("layercam.npz", len(out))
    # classic CAM (fc-weight CAM for every class) - AlternatingDirectionCutLoss.py:320-403
    cam_cls = lift(f"{REF}/AlternatingDirectionCutLoss.py", {"CAMGenerator"})["CAMGenerator"]
    cg = cam_cls(net)
    co = {"image": imgs[1].numpy()}
    co["all_cams"] = cg.generate_all_cams(imgs[1]).numpy()                      # (7,14,14)
    m_bg, max_obj = cg.generate_bg_cam(imgs[1], [1, 4], alpha=2.0)
    co["m_bg"], co["max_obj"] = m_bg.numpy(), max_obj.numpy()                   # (224,224) each
    np.savez_compressed(f"{HERE}/classic_cam.npz", **co)
    print("classic_cam.npz", len(co))


def gen_layercam_bg():
    """``generate_bg_cam`` of the notebook LayerCAMGenerator (AlternatingDirectionCutLoss.py:296-318) on the fixture net."""
    nb = lift(f"{REF}/AlternatingDirectionCutLoss.py", {"LayerCAMGenerator"})["LayerCAMGenerator"]
    torch.manual_seed(7)
    net = ToyCAMNet()
    g = torch.Generator().manual_seed(8)
    imgs = torch.rand(2, 3, 112, 112, generator=g)
    out = {"state/" + k: v.numpy() for k, v in net.state_dict().items()}
    out["image"] = imgs[1].numpy()
    out["class_idx"] = np.array([4])
    gen = nb(net, ["layer3", "layer4"])
    out["all_cams"] = gen.generate(imgs[1].clone(), torch.tensor([4])).numpy()           # (1,224,224), alpha 1.0
    for n in ("layer3", "layer4"):
        out[f"act_{n}"] = gen.activations[n].detach().numpy()
        out[f"grad_{n}"] = gen.gradients[n].detach().numpy()
    for alpha in (2.0, 0.5):
        m_bg, max_obj = gen.generate_bg_cam(imgs[1].clone(), torch.tensor([4]), alpha=alpha)
        out[f"m_bg_a{alpha}"], out[f"max_obj_a{alpha}"] = m_bg.numpy(), max_obj.numpy()   # (224,224) each
    np.savez_compressed(f"{HERE}/layercam_bg.npz", **out)
    print("layercam_bg.npz", len(out))


def gen_layercam_wide():
    """Channel counts that exercise every level of torch's cascade summation (300 = 256 + 2 x 16 + 12 left over, 600 = 2 x 256 +
    5 x 16 + 8) on 14 x 14 maps (192 vectorised pixels + 4 scalar-column pixels): the reference's own bodies pin the ORDER
    of the channel sum the HIP epilogue reproduces bit for bit."""
    mod = lift(f"{REF}/LayerCAM.py", {"LayerCAMGenerator"})["LayerCAMGenerator"]
    nb = lift(f"{REF}/AlternatingDirectionCutLoss.py", {"LayerCAMGenerator"})["LayerCAMGenerator"]
    torch.manual_seed(17)
    net = ToyCAMNet(c3=300, c4=600)
    g = torch.Generator().manual_seed(18)
    img = torch.rand(3, 112, 112, generator=g)
    ci = torch.tensor([3])
    out = {}
    gen = mod(net, ["layer3", "layer4"])
    out["modular_cam_a1.0"] = gen.generate(img, 1.0, class_idx=ci).numpy()
    for n in ("layer3", "layer4"):
        out[f"act_{n}"] = gen.activations[n].detach().numpy()
        out[f"grad_{n}"] = gen.gradients[n].detach().numpy()
    out["modular_cam_a3.0"] = gen.generate(img, 3.0, class_idx=ci).numpy()
    gen = nb(net, ["layer3", "layer4"])
    out["notebook_cam_a0.5"] = gen.generate(img, class_idx=ci, alpha=0.5).numpy()
    assert np.array_equal(gen.activations["layer4"].detach().numpy(), out["act_layer4"])
    np.savez_compressed(f"{HERE}/layercam_wide.npz", **out)
    print("layercam_wide.npz", len(out))


def gen_refine_and_metrics():
    ns = lift(f"{REF}/AlternatingDirectionCutLoss.py", {"LocalNormalizedCutLoss", "refine_pseudo_mask"})
    met = lift(f"{REF}/ExtraUtilities.py", {"compute_iou_and_acc"})["compute_iou_and_acc"]
    H = W = 32
    g = torch.Generator().manual_seed(31)
    img = smooth_image(1, H, W, 32)[0]
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    blob = (((yy - 15) ** 2 + (xx - 14) ** 2) < 81).float()
    logits = torch.stack([1.5 - 3 * blob, 3 * blob - 1.5]) + 0.3 * torch.randn(2, H, W, generator=g)
    mask = (((yy - 13) ** 2 + (xx - 17) ** 2) < 64).long() * 255

    class Stub(nn.Module):
        def __init__(self):
            super().__init__()
            self.p = nn.Parameter(torch.zeros(1))

        def forward(self, x):
            return {"out": logits.unsqueeze(0) + 0 * self.p}

    out = {"image": img.numpy(), "logits": logits.numpy(), "mask": mask.numpy()}
    refined = ns["refine_pseudo_mask"](Stub(), img, mask, threshold=0.3, lr=1e-4, num_steps=10,
                                       lambda_boundary=0.1)
    out["refined_callsite"] = refined.numpy()                 # call-site hyper-parameters (:806)
    refined = ns["refine_pseudo_mask"](Stub(), img, mask)      # defaults: lr 1e-2, 20 steps, thr .5
    out["refined_default"] = refined.numpy()
    refined = ns["refine_pseudo_mask"](Stub(), img, mask, lr=0.5, num_steps=12, threshold=0.5)
    out["refined_lr0.5"] = refined.numpy()                     # large lr so the mask really moves
    pm = (torch.rand(24, 24, generator=g) > 0.5).long()
    tm = (torch.rand(24, 24, generator=g) > 0.4).long()
    iou, acc = met(pm, tm)
    out["metric_pred"], out["metric_true"] = pm.numpy(), tm.numpy()
    out["metric_iou_acc"] = np.array([iou, acc])
    np.savez_compressed(f"{HERE}/refine_metrics.npz", **out)
    print("refine_metrics.npz", len(out))


_KEEP_LARGEST_PY39 = r'''
import ast, sys, numpy as np
from skimage.measure import label as lb, regionprops
src = open("/root/reference/TraditionalModel/PsuedoMasks.py").read()
node = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "keep_largest"]
ns = {"lb": lb, "regionprops": regionprops, "np": np}
exec(compile(ast.Module(body=node, type_ignores=[]), "PsuedoMasks.py", "exec"), ns)
rng = np.random.RandomState(5)
cases = {}
m = np.zeros((8, 8), np.uint8); m[0:2, 0:2] = 1; m[2:5, 2:5] = 1; m[6, 7] = 1        # diagonal touch
cases["diag"] = m
m = np.zeros((10, 10), np.uint8); m[1:3, 1:4] = 1; m[6:8, 5:8] = 1                   # equal areas
cases["tie"] = m
cases["tie_flipped"] = m[::-1, ::-1].copy()
cases["empty"] = np.zeros((6, 9), np.uint8)
cases["full"] = np.ones((5, 7), np.uint8)
cases["rand224"] = (rng.rand(224, 224) < 0.45).astype(np.uint8)
cases["rand64_sparse"] = (rng.rand(64, 64) < 0.2).astype(np.uint8)
out = {}
for k, v in cases.items():
    out["in_" + k] = v
    out["out_" + k] = np.asarray(ns["keep_largest"](v)).astype(np.uint8)
np.savez_compressed(sys.argv[1], **out)
print("keep_largest.npz", len(out))
'''


def gen_lovasz():
    """lovasz_softmax of the reference's own file (LossFunctions/Lovasz-Softmax_Loss.py: torch + numpy only; its
    ``Variable`` is the identity on modern torch): loss and d loss / d probas for softmax probabilities of seeded logits."""
    from torch.autograd import Variable
    try:
        from itertools import ifilterfalse
    except ImportError:
        from itertools import filterfalse as ifilterfalse
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")                       # ``classes is 'present'``: SyntaxWarning
        ns = lift(f"{REF}/LossFunctions/Lovasz-Softmax_Loss.py",
                  {"lovasz_grad", "lovasz_softmax", "lovasz_softmax_flat", "flatten_probas", "mean", "isnan"},
                  {"Variable": Variable, "ifilterfalse": ifilterfalse})
    out, meta = {}, []
    cases = [(2, 2, 16, 16, "present", False, None), (3, 2, 12, 20, "present", False, None), (2, 3, 8, 8, "present", False, None),
             (2, 3, 8, 8, "all", False, None), (2, 2, 16, 16, "present", True, None), (2, 4, 10, 6, "present", False, 3)]
    for i, (B, C, H, W, classes, per_image, ignore) in enumerate(cases):
        g = torch.Generator().manual_seed(500 + i)
        logits = 1.5 * torch.randn(B, C, H, W, generator=g)
        labels = torch.randint(0, C, (B, H, W), generator=g)
        if i == 2:
            labels[labels == 2] = 0                                # class 2 absent: 'present' skips it
        probas = F.softmax(logits, dim=1).detach().requires_grad_()
        loss = ns["lovasz_softmax"](probas, labels, classes=classes, per_image=per_image, ignore=ignore)
        loss.backward()
        out[f"lov{i}_probas"], out[f"lov{i}_labels"] = probas.detach().numpy(), labels.numpy()
        out[f"lov{i}_loss"], out[f"lov{i}_grad"] = np.float64(loss.item()), probas.grad.numpy()
        meta.append(dict(B=B, C=C, H=H, W=W, classes=classes, per_image=per_image, ignore=ignore))
    out["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(f"{HERE}/lovasz.npz", **out)
    print("lovasz.npz", len(out))


def gen_bottleneck():
    """The only network code the reference holds itself: ``Bottleneck`` vendored from an old torchvision in
    PretrainedBasnetModel/model/resnet_model.py:99-135 (1x1 -> 3x3 with the stride on it, i.e. the v1.5 wiring ->
    1x1 x4, residual add, ReLU after the add; no dilation argument).  Train-mode forward + backward of three blocks
    - identity shortcut; projection shortcut at stride 1 (layer1.0's form); projection shortcut at stride 2
    (layer2.0's form) - pin ``oracle.models.Bottleneck(dilation=1)`` and the HIP ``nn.Bottleneck``: outputs, input
    gradient, every parameter gradient and the BatchNorm running statistics after the forward."""
    Block = lift("/root/reference/PretrainedBasnetModel/model/resnet_model.py", {"Bottleneck"})["Bottleneck"]
    out, meta = {}, []
    cases = [(64, 16, 1, False, 2, 12, 12), (32, 16, 1, True, 2, 10, 14), (32, 16, 2, True, 3, 12, 10)]
    for i, (inpl, planes, stride, down, B, H, W) in enumerate(cases):
        torch.manual_seed(900 + i)
        ds = None
        if down:
            ds = nn.Sequential(nn.Conv2d(inpl, planes * 4, kernel_size=1, stride=stride, bias=False),
                               nn.BatchNorm2d(planes * 4))
        blk = Block(inpl, planes, stride, ds)
        g = torch.Generator().manual_seed(910 + i)
        for m in blk.modules():
            if isinstance(m, nn.BatchNorm2d):               # non-trivial affine / running statistics
                m.weight.data.copy_(torch.rand(m.num_features, generator=g) * 0.5 + 0.75)
                m.bias.data.copy_(torch.randn(m.num_features, generator=g) * 0.1)
                m.running_mean.copy_(torch.randn(m.num_features, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.num_features, generator=g) + 0.5)
        for k, v in blk.state_dict().items():
            out[f"b{i}/state/{k}"] = v.detach().clone().numpy()
        x = torch.randn(B, inpl, H, W, generator=g).requires_grad_()
        dy = torch.randn(B, planes * 4, (H - 1) // stride + 1, (W - 1) // stride + 1, generator=g)
        blk.train()
        y = blk(x)
        y.backward(dy)
        out[f"b{i}/x"], out[f"b{i}/dy"] = x.detach().numpy(), dy.numpy()
        out[f"b{i}/y"], out[f"b{i}/dx"] = y.detach().numpy(), x.grad.numpy()
        for k, p in blk.named_parameters():
            out[f"b{i}/grad/{k}"] = p.grad.numpy()
        for k, v in blk.state_dict().items():
            if "running" in k:
                out[f"b{i}/after/{k}"] = v.detach().clone().numpy()
        blk.eval()
        out[f"b{i}/y_eval"] = blk(x.detach()).detach().numpy()
        meta.append(dict(inplanes=inpl, planes=planes, stride=stride, downsample=down))
    out["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(f"{HERE}/bottleneck.npz", **out)
    print("bottleneck.npz", len(out))


def gen_eval_helpers():
    """``evaluate_classification`` (ClassificationModel.py:109-150) and ``evaluate_layercam_on_test_set``
    (LayerCAM.py:84-130) run on stubs: a "model" that returns stored logits and a "generator" that returns stored
    CAMs - the fixture pins the metric procedure (accuracy / macro-F1 formulas; threshold, trimap binarisation,
    nearest resize, the 11-image cap), not a network.  ``.cuda()`` in the reference body is made the identity."""
    import contextlib
    import io
    met = lift(f"{REF}/ExtraUtilities.py", {"compute_iou_and_acc"})["compute_iou_and_acc"]
    ec = lift(f"{REF}/ClassificationModel.py", {"evaluate_classification"})["evaluate_classification"]
    ev = lift(f"{REF}/LayerCAM.py", {"evaluate_layercam_on_test_set"}, {"compute_iou_and_acc": met})["evaluate_layercam_on_test_set"]
    g = torch.Generator().manual_seed(77)
    out = {}
    nb, bs, nc = 5, 8, 37
    logits = torch.randn(nb, bs, nc, generator=g)
    labels = torch.randint(0, nc, (nb, bs), generator=g)
    logits[torch.arange(nb)[:, None], torch.arange(bs)[None, :], labels] += 2.0 * (torch.rand(nb, bs, generator=g) > 0.4)

    class Stub(nn.Module):
        def __init__(self):
            super().__init__()
            self.k = 0

        def forward(self, x):
            k, self.k = self.k, self.k + 1
            return logits[k], None

    loader = [(torch.zeros(bs, 1), (labels[k], None)) for k in range(nb)]
    with contextlib.redirect_stdout(io.StringIO()):
        acc, f1 = ec(Stub(), loader, torch.device("cpu"), num_classes=nc)
    out["cls_logits"], out["cls_labels"], out["cls_acc_f1"] = logits.numpy(), labels.numpy(), np.array([acc, f1])

    n = 13                                                   # the reference stops after 11 (i >= 10)
    cams = smooth_image(n, 56, 56, 78)[:, 0]                 # (n,56,56) in [0,1]
    tri = torch.randint(1, 4, (n, 1, 1, 56, 56), generator=g)          # trimap values 1..3, foreground = 1
    tri_small = torch.randint(1, 4, (n, 1, 1, 40, 50), generator=g)    # every third image: another size -> nearest resize
    lab = torch.randint(0, nc, (n,), generator=g)

    class Gen:
        def __init__(self):
            self.k = 0

        def generate(self, img, class_idx=None, alpha=1.0):
            k, self.k = self.k, self.k + 1
            return cams[k:k + 1].clone()

    test_loader = [(torch.zeros(1, 3, 56, 56), (lab[k:k + 1], (tri_small if k % 3 == 2 else tri)[k])) for k in range(n)]
    orig = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            res = ev(Gen(), test_loader, alpha=1.0, cam_thresh=0.3)
    finally:
        torch.Tensor.cuda = orig
    out["cam_maps"] = cams.numpy().astype(np.float16).astype(np.float32)      # stored rounded: the stub returns these exact values
    out["cam_tri"], out["cam_tri_small"], out["cam_labels"] = tri.numpy().astype(np.uint8), tri_small.numpy().astype(np.uint8), lab.numpy()
    # re-run on the rounded maps so that fixture inputs and outputs correspond exactly
    cams = torch.from_numpy(out["cam_maps"])
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            res = ev(Gen(), test_loader, alpha=1.0, cam_thresh=0.3)
    finally:
        torch.Tensor.cuda = orig
    out["cam_iou_acc"] = np.array([res["layercam_fg_iou"], res["layercam_fg_acc"]])
    np.savez_compressed(f"{HERE}/eval_helpers.npz", **out)
    print("eval_helpers.npz", len(out), out["cls_acc_f1"], out["cam_iou_acc"])


def gen_keep_largest():
    py39 = "/opt/conda/bin/python3.9"
    subprocess.run([py39, "-c", _KEEP_LARGEST_PY39, f"{HERE}/keep_largest.npz"], check=True)


if __name__ == "__main__":
    torch.set_num_threads(4)
    gens = dict(losses=gen_losses, layercam=gen_layercam, layercam_bg=gen_layercam_bg, layercam_wide=gen_layercam_wide, refine=gen_refine_and_metrics, lovasz=gen_lovasz,
                bottleneck=gen_bottleneck, eval_helpers=gen_eval_helpers, keep_largest=gen_keep_largest)
    for name in (sys.argv[1:] or list(gens)):        # python make_golden.py [losses layercam ...]: only those
        gens[name]()
