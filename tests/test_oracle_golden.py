"""Pin the CPU oracle against golden vectors captured from the reference's own function bodies
(tests/golden/make_golden.py).  CPU only."""
import json

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

import oracle

T = torch.from_numpy
REL = 1e-4   # oracle vs reference: same fp32 arithmetic, different association


def _close(a, b, rel=REL, abs_=1e-7):
    a, b = torch.as_tensor(a), torch.as_tensor(b)
    scale = b.abs().max().item()
    assert (a - b).abs().max().item() <= rel * scale + abs_, ((a - b).abs().max().item(), scale)


def test_ncut_loss_and_grad(golden):
    g = golden("losses")
    meta = [m for m in json.loads(str(g["meta"])) if m["kind"] == "ncut"]
    assert len(meta) == 4
    for m in meta:
        i = m["idx"]
        preds = T(g[f"ncut{i}_preds"]).requires_grad_()
        loss = oracle.LocalNormalizedCutLoss(m["sigma_color"], m["window"])(preds, T(g[f"ncut{i}_image"]))
        loss.backward()
        _close(loss.detach(), g[f"ncut{i}_loss"])
        _close(preds.grad, g[f"ncut{i}_grad"])


def test_ncut_3d_path(golden):
    g = golden("losses")
    preds = T(g["ncut3d_preds"]).requires_grad_()
    loss = oracle.LocalNormalizedCutLoss(0.1, 5)(preds, T(g["ncut3d_image"]))
    loss.backward()
    _close(loss.detach(), g["ncut3d_loss"])
    _close(preds.grad, g["ncut3d_grad"])


def test_compute_affinities(golden):
    g = golden("losses")
    aff = oracle.compute_affinities(T(g["aff_image"]), 0.1, 5, 5)
    assert len(aff) == 24 and aff[0].shape == (2, 1, 16, 16)
    _close(torch.stack(aff), g["aff_maps"])
    single = oracle.ConstrainToBoundaryLossSingle.compute_affinities_single(T(g["aff_image"])[1], 0.1, 5, 5)
    assert len(single) == 24 and single[0].shape == (1, 16, 16)
    _close(torch.stack(single)[:, 0], g["aff_maps"][:, 1, 0])


def test_boundary_loss_and_grad(golden):
    g = golden("losses")
    meta = [m for m in json.loads(str(g["meta"])) if m["kind"] == "boundary"]
    assert len(meta) == 3
    for m in meta:
        i = m["idx"]
        p = T(g[f"bnd{i}_preds"]).requires_grad_()
        loss = oracle.ConstrainToBoundaryLossSingle(m["sigma_color"], m["sigma_space"], m["window"])(
            p, T(g[f"bnd{i}_image"]))
        loss.backward()
        _close(loss.detach(), g[f"bnd{i}_loss"])
        _close(p.grad, g[f"bnd{i}_grad"])


def test_layercam_epilogue_both_variants(golden):
    """The epilogue is where an INTEGER output (the pseudo mask) is decided: the oracle equals the reference's bodies bit
    for bit on identical activations / gradients (same torch ops in the same order), not merely within a tolerance."""
    g = golden("layercam")
    for i in range(2):
        acts = [T(g[f"act_layer3_{i}"]), T(g[f"act_layer4_{i}"])]
        grads = [T(g[f"grad_layer3_{i}"]), T(g[f"grad_layer4_{i}"])]
        for a in (0.5, 1.0, 2.0):
            assert torch.equal(oracle.layercam_epilogue(acts, grads, (224, 224), a, "modular"), T(g[f"modular_cam_{i}_a{a}"]))
        for a in (0.5, 2.0):
            assert torch.equal(oracle.layercam_epilogue(acts, grads, (224, 224), a, "notebook"), T(g[f"notebook_cam_{i}_a{a}"]))
    g = golden("layercam_wide")        # 300 / 600 channels: every level of torch's cascade summation
    acts, grads = [T(g["act_layer3"]), T(g["act_layer4"])], [T(g["grad_layer3"]), T(g["grad_layer4"])]
    assert torch.equal(oracle.layercam_epilogue(acts, grads, (224, 224), 1.0, "modular"), T(g["modular_cam_a1.0"]))
    assert torch.equal(oracle.layercam_epilogue(acts, grads, (224, 224), 3.0, "modular"), T(g["modular_cam_a3.0"]))
    assert torch.equal(oracle.layercam_epilogue(acts, grads, (224, 224), 0.5, "notebook"), T(g["notebook_cam_a0.5"]))


def _cascade_rows(x):
    """ATen SumKernel.cpp multi_row_sum over axis 0 of x (n, P): four levels, level step 16 (n <= 2^19)."""
    import numpy as np
    f32 = np.float32
    acc = [np.zeros(x.shape[1], f32) for _ in range(4)]
    i, n = 0, x.shape[0]
    while i + 16 <= n:
        for _ in range(16):
            acc[0] = acc[0] + x[i]
            i += 1
        for j in range(1, 4):
            acc[j] = acc[j] + acc[j - 1]
            acc[j - 1] = np.zeros_like(acc[0])
            if i & (15 << (4 * j)):
                break
    while i < n:
        acc[0] = acc[0] + x[i]
        i += 1
    for j in range(1, 4):
        acc[0] = acc[0] + acc[j]
    return acc[0]


def test_channel_sum_order_the_hip_epilogue_reproduces():
    """csrc/layercam_optim.hip states torch-CPU's order of `relu(g * a).sum(dim=1)` on a contiguous NCHW tensor: the
    four-level cascade for pixels below hw - hw % 32, four interleaved streams for the rest.  This is that statement in
    numpy, checked against the torch of this machine - if a torch release changes the order, this test says so before
    the GPU tests do.  The reference's OWN result depends on the host's thread count: when the last thread of ATen's
    column split is left with fewer than 32 columns (16 or 64 threads on a 14 x 14 map) those go through the cascade too
    (`layercam_tail_mod = 0` in the library); 1-12, 24 and 32 threads give the order of the fixtures (the default)."""
    import numpy as np
    from conftest import cpu_threads

    def expected(X, tail_mod):
        C, hw = X.shape
        out = _cascade_rows(X)
        for p in range(hw - hw % tail_mod if tail_mod else hw, hw):
            n = C // 4
            ps = _cascade_rows(X[:4 * n, p].reshape(n, 4))
            for c in range(4 * n, C):
                ps[0] = ps[0] + X[c, p]
            out[p] = ((ps[0] + ps[1]) + ps[2]) + ps[3]
        return out

    g = torch.Generator().manual_seed(3)
    for threads in (1, 4, 8):
        with cpu_threads(threads):
            for C, hw in ((1024, 196), (2048, 196), (300, 196), (37, 169), (600, 81), (4097, 25), (64, 784), (18, 100), (530, 121)):
                for B in (1, 2):
                    w = F.relu(torch.randn(B, C, hw, 1, generator=g) * torch.randn(B, C, hw, 1, generator=g))
                    ref = w.sum(dim=1).numpy().reshape(B, hw)
                    for b in range(B):
                        mine = expected(w[b].numpy().reshape(C, hw), 32)
                        assert np.array_equal(mine, ref[b]), (threads, C, hw, B, b, int((mine != ref[b]).sum()))
    with cpu_threads(16):       # the GPU boxes' host: 16 threads
        for C in (1024, 2048):
            w = F.relu(torch.randn(1, C, 14, 14, generator=g) * torch.randn(1, C, 14, 14, generator=g))
            assert np.array_equal(expected(w[0].numpy().reshape(C, 196), 0), w.sum(dim=1).numpy().reshape(196))


class _Toy(nn.Module):
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


def test_layercam_generator_hooks(golden):
    g = golden("layercam")
    net = _Toy()
    net.load_state_dict({k[6:]: T(g[k]) for k in g.files if k.startswith("state/")})
    imgs = T(g["images"])
    for variant in ("modular", "notebook"):
        gen = oracle.LayerCAMGenerator(net, ["layer3", "layer4"], variant=variant)
        for i in range(2):
            ci = torch.tensor([int(g["class_idx"][i])])
            cam = gen.generate(imgs[i], alpha=2.0, class_idx=ci)
            assert cam.shape == (1, 224, 224)
            _close(cam, g[f"{variant}_cam_{i}_a2.0"], rel=1e-4)
        _close(gen.generate(imgs[0]), g[f"{variant}_cam_argmax"], rel=1e-4)   # argmax default
    # notebook keyword order generate(images, class_idx, alpha) is accepted too
    gen = oracle.LayerCAMGenerator(net, ["layer3", "layer4"], variant="notebook")
    _close(gen.generate(imgs[1], torch.tensor([4]), 0.5), g["notebook_cam_1_a0.5"], rel=1e-4)
    _close(gen(imgs[1], class_idx=torch.tensor([4]), alpha=0.5), g["notebook_cam_1_a0.5"], rel=1e-4)


def test_layercam_generate_bg_cam(golden):
    """``generate_bg_cam`` of the notebook LayerCAMGenerator (reference AlternatingDirectionCutLoss.py:296-318): the oracle's
    restatement against the vectors the reference's own body produced."""
    g = golden("layercam_bg")
    net = _Toy()
    net.load_state_dict({k[6:]: T(g[k]) for k in g.files if k.startswith("state/")})
    gen = oracle.LayerCAMGenerator(net, ["layer3", "layer4"], variant="notebook")
    img, ci = T(g["image"]), torch.tensor(g["class_idx"])
    _close(gen.generate(img, ci), g["all_cams"], rel=1e-5)                     # notebook order (image, class_idx), alpha 1.0
    for alpha in (2.0, 0.5):
        m_bg, max_obj = gen.generate_bg_cam(img, ci, alpha=alpha)
        assert tuple(m_bg.shape) == tuple(max_obj.shape) == (224, 224)
        _close(m_bg, g[f"m_bg_a{alpha}"], rel=1e-5)
        _close(max_obj, g[f"max_obj_a{alpha}"], rel=1e-5)


def test_classic_cam(golden):
    g, gl = golden("classic_cam"), golden("layercam")
    net = _Toy()
    net.load_state_dict({k[6:]: T(gl[k]) for k in gl.files if k.startswith("state/")})
    cg = oracle.CAMGenerator(net)
    _close(cg.generate_all_cams(T(g["image"])), g["all_cams"], rel=1e-5)
    m_bg, max_obj = cg.generate_bg_cam(T(g["image"]), [1, 4], alpha=2.0)
    _close(m_bg, g["m_bg"], rel=1e-5)
    _close(max_obj, g["max_obj"], rel=1e-5)


def test_keep_largest_bit_exact(golden):
    g = golden("keep_largest")
    names = [k[3:] for k in g.files if k.startswith("in_")]
    assert {"diag", "tie", "tie_flipped", "empty", "full", "rand224"} <= set(names)
    for n in names:
        out = oracle.keep_largest(g["in_" + n])
        assert out.dtype == np.uint8 and np.array_equal(out, g["out_" + n]), n


def test_refine_pseudo_mask(golden):
    g = golden("refine_metrics")
    logits = T(g["logits"])

    class Stub(nn.Module):
        def forward(self, x):
            return {"out": logits.unsqueeze(0)}

    img, mask = T(g["image"]), T(g["mask"])
    r = oracle.refine_pseudo_mask(Stub(), img, mask, threshold=0.3, lr=1e-4, num_steps=10)
    assert np.array_equal(r.numpy(), g["refined_callsite"])
    r = oracle.refine_pseudo_mask(Stub(), img, mask)
    assert np.array_equal(r.numpy(), g["refined_default"])
    r = oracle.refine_pseudo_mask(Stub(), img, mask, lr=0.5, num_steps=12, threshold=0.5)
    assert (r.numpy() != g["refined_lr0.5"]).mean() <= 0.002      # large-lr case: near-threshold pixels
    assert not np.array_equal(g["refined_lr0.5"], (g["mask"] == 255).astype(np.float32))


def test_metrics(golden):
    g = golden("refine_metrics")
    iou, acc = oracle.compute_iou_and_acc(T(g["metric_pred"]), T(g["metric_true"]))
    assert abs(iou - g["metric_iou_acc"][0]) < 1e-12 and abs(acc - g["metric_iou_acc"][1]) < 1e-12


def test_model_structure_selfchecks():
    """torchvision is absent: pin the restated architectures by their published parameter counts."""
    n = lambda m: sum(p.numel() for p in m.parameters())
    assert n(oracle.ResNet50Trunk()) == 25_557_032
    assert n(oracle.DeepLabV3ResNet50(21, True)) == 42_004_074
    seg = oracle.build_segmentation_model()
    assert n(seg) == 41_999_191
    assert sum(p.numel() for k, p in seg.named_parameters() if not k.startswith("aux_")) == 39_633_986
    cam = oracle.FrozenResNetCAM()
    assert n(cam) == 23_583_845
    assert [k for k, p in cam.named_parameters() if p.requires_grad] == ["fc.weight", "fc.bias"]
    keys = set(seg.state_dict())
    for k in ("backbone.layer4.2.conv3.weight", "classifier.0.convs.4.1.weight", "classifier.0.project.1.running_var",
              "classifier.4.bias", "aux_classifier.4.weight", "backbone.layer2.0.downsample.1.weight"):
        assert k in keys, k
    assert {"layer0.0.weight", "layer0.1.running_mean", "layer4.0.downsample.0.weight", "fc.bias"} <= set(cam.state_dict())


def test_model_shapes_small():
    torch.manual_seed(0)
    seg = oracle.build_segmentation_model().eval()
    with torch.no_grad():
        out = seg(torch.randn(1, 3, 64, 64))
    assert out["out"].shape == (1, 2, 64, 64) and out["aux"].shape == (1, 21, 64, 64)
    cam = oracle.FrozenResNetCAM().eval()
    with torch.no_grad():
        logits, feats = cam(torch.randn(1, 3, 64, 64))
    assert logits.shape == (1, 37)
    assert [tuple(f.shape[1:]) for f in feats] == [(512, 8, 8), (1024, 4, 4), (2048, 4, 4)]


def _run_trunk(t, x):
    x = t.maxpool(t.relu(t.bn1(t.conv1(x))))
    x = t.layer4(t.layer3(t.layer2(t.layer1(x))))
    return t.fc(t.avgpool(x).flatten(1))


def _conv_macs(model, size, run=None):
    """(total MACs, {name: (Cin, Cout, k, stride, dilation, Hout)}) of every nn.Conv2d / nn.Linear, by shape inference on
    the meta device (no arithmetic)."""
    macs, table, hooks = [0], {}, []
    names = {m: k for k, m in model.named_modules()}

    def hook(m, inp, out):
        if isinstance(m, nn.Conv2d):
            macs[0] += out.numel() // out.shape[0] * (m.in_channels // m.groups) * m.kernel_size[0] * m.kernel_size[1]
            table[names[m]] = (m.in_channels, m.out_channels, m.kernel_size[0], m.stride[0], m.dilation[0], out.shape[-1])
        else:
            macs[0] += m.in_features * m.out_features
    for m in model.modules():
        if isinstance(m, (nn.Conv2d, nn.Linear)):
            hooks.append(m.register_forward_hook(hook))
    model = model.to("meta").eval()
    with torch.no_grad():
        x = torch.empty(1, 3, size, size, device="meta")
        run(model, x) if run is not None else model(x)
    for h in hooks:
        h.remove()
    return macs[0], table


def test_model_wiring_is_pinned_by_published_op_counts_and_the_shape_table():
    """Parameter counts cannot see a misplaced stride or dilation; multiply-accumulate counts and per-layer output sizes
    can.  torchvision's model cards: resnet50 4.09 GMACs @224, deeplabv3_resnet50 (21 classes, aux head) 178.72 GMACs
    @520; the per-layer table is SURVEY.md 8a (256 x 256 input)."""
    macs, _ = _conv_macs(oracle.ResNet50Trunk(), 224, _run_trunk)
    assert abs(macs / 1e9 - 4.089) < 0.005, macs
    macs, _ = _conv_macs(oracle.DeepLabV3ResNet50(21, True), 520)
    assert abs(macs / 1e9 - 178.72) < 0.05, macs
    macs, table = _conv_macs(oracle.build_segmentation_model(), 256)
    assert abs(macs / 1e9 - 43.31) < 0.02, macs                       # 43.31 GMACs with the aux head
    want = {  # name: (Cin, Cout, k, stride, dilation, Hout)
        "backbone.conv1": (3, 64, 7, 2, 1, 128),
        "backbone.layer1.0.conv2": (64, 64, 3, 1, 1, 64),
        "backbone.layer2.0.conv2": (128, 128, 3, 2, 1, 32),
        "backbone.layer2.0.downsample.0": (256, 512, 1, 2, 1, 32),
        "backbone.layer3.0.conv2": (256, 256, 3, 1, 1, 32),          # stride replaced by dilation: block 0 keeps d = 1
        "backbone.layer3.1.conv2": (256, 256, 3, 1, 2, 32),
        "backbone.layer3.5.conv2": (256, 256, 3, 1, 2, 32),
        "backbone.layer3.0.downsample.0": (512, 1024, 1, 1, 1, 32),
        "backbone.layer4.0.conv2": (512, 512, 3, 1, 2, 32),
        "backbone.layer4.1.conv2": (512, 512, 3, 1, 4, 32),
        "backbone.layer4.2.conv3": (512, 2048, 1, 1, 1, 32),
        "classifier.0.convs.0.0": (2048, 256, 1, 1, 1, 32),
        "classifier.0.convs.1.0": (2048, 256, 3, 1, 12, 32),
        "classifier.0.convs.2.0": (2048, 256, 3, 1, 24, 32),
        "classifier.0.convs.3.0": (2048, 256, 3, 1, 36, 32),
        "classifier.0.convs.4.1": (2048, 256, 1, 1, 1, 1),
        "classifier.0.project.0": (1280, 256, 1, 1, 1, 32),
        "classifier.1": (256, 256, 3, 1, 1, 32),
        "classifier.4": (256, 2, 1, 1, 1, 32),
        "aux_classifier.0": (1024, 256, 3, 1, 1, 32),
        "aux_classifier.4": (256, 21, 1, 1, 1, 32),
    }
    for k, v in want.items():
        assert table[k] == v, (k, table[k], v)
    assert len(table) == 63                                            # 63 convolutions
    # FrozenResNetCAM: layer4 dilated (stride-16 features), layer3 strided
    _, t = _conv_macs(oracle.FrozenResNetCAM(37), 224)
    assert t["layer3.0.conv2"] == (256, 256, 3, 2, 1, 14) and t["layer4.0.conv2"] == (512, 512, 3, 1, 1, 14)
    assert t["layer4.1.conv2"] == (512, 512, 3, 1, 2, 14) and t["layer0.0"] == (3, 64, 7, 2, 1, 112)


def test_product_models_are_wired_like_the_oracle():
    """The HIP path's modules (weaklysuperviseddl_amd.nn) carry the same (in, out, kernel, stride, padding, dilation) per
    state_dict name as the oracle's torch.nn modules - construction only, no kernel runs."""
    from weaklysuperviseddl_amd import nn as wnn
    from weaklysuperviseddl_amd.TraditionalModel import FrozenResNetCAM, build_segmentation_model

    def geom_ref(m):
        return {k: (c.in_channels, c.out_channels, c.kernel_size[0], c.stride[0], c.padding[0], c.dilation[0], c.bias is not None)
                for k, c in m.named_modules() if isinstance(c, nn.Conv2d)}

    def geom_mine(m):
        return {k: (c.in_channels, c.out_channels, c.kernel_size, c.stride, c.padding, c.dilation, c.bias is not None)
                for k, c in m.named_modules() if isinstance(c, wnn.Conv2d)}
    assert geom_mine(build_segmentation_model()) == geom_ref(oracle.build_segmentation_model())
    assert geom_mine(FrozenResNetCAM(37)) == geom_ref(oracle.FrozenResNetCAM(37))


def test_lovasz_softmax_loss_and_grad(golden):
    """The oracle's Lovasz-softmax against the reference's own function bodies (LossFunctions/Lovasz-Softmax_Loss.py):
    classes='present' (a class without pixels skipped) / 'all', per_image, an ignored label."""
    g = golden("lovasz")
    meta = json.loads(str(g["meta"]))
    assert len(meta) == 6
    for i, m in enumerate(meta):
        p = T(g[f"lov{i}_probas"]).requires_grad_()
        loss = oracle.lovasz_softmax(p, T(g[f"lov{i}_labels"]), classes=m["classes"], per_image=m["per_image"], ignore=m["ignore"])
        loss.backward()
        _close(loss.detach(), g[f"lov{i}_loss"], rel=1e-5)
        _close(p.grad, g[f"lov{i}_grad"], rel=1e-5)


def _bottleneck_from_fixture(g, i, meta, make_block, make_conv, make_bn, make_seq):
    m = meta[i]
    ds = None
    if m["downsample"]:
        ds = make_seq(make_conv(m["inplanes"], m["planes"] * 4, 1, m["stride"]), make_bn(m["planes"] * 4))
    blk = make_block(m["inplanes"], m["planes"], m["stride"], 1, ds)
    pre = f"b{i}/state/"
    blk.load_state_dict({k[len(pre):]: T(g[k]).clone() for k in g.files if k.startswith(pre)})
    return blk


def test_bottleneck_matches_the_references_vendored_block(golden):
    """The one piece of network code the reference holds itself - ``Bottleneck`` in
    PretrainedBasnetModel/model/resnet_model.py:99-135 (vendored from torchvision; stride on the 3x3 = the v1.5 wiring
    torchvision's resnet50 / deeplabv3_resnet50 use, no dilation argument) - pins ``oracle.models.Bottleneck`` at
    dilation 1: train-mode forward, input gradient, every parameter gradient, the running statistics written by the
    forward, and the eval-mode forward, for the identity shortcut and both projection-shortcut forms."""
    g = golden("bottleneck")
    meta = json.loads(str(g["meta"]))
    assert [(m["stride"], m["downsample"]) for m in meta] == [(1, False), (1, True), (2, True)]
    for i in range(len(meta)):
        blk = _bottleneck_from_fixture(
            g, i, meta, oracle.models.Bottleneck,
            lambda ci, co, k, s: nn.Conv2d(ci, co, k, stride=s, bias=False), nn.BatchNorm2d, nn.Sequential).train()
        x = T(g[f"b{i}/x"]).clone().requires_grad_()
        y = blk(x)
        y.backward(T(g[f"b{i}/dy"]))
        _close(y.detach(), g[f"b{i}/y"], rel=1e-5)
        _close(x.grad, g[f"b{i}/dx"], rel=1e-5)
        for k, p in blk.named_parameters():
            _close(p.grad, g[f"b{i}/grad/{k}"], rel=1e-5)
        for k, v in blk.state_dict().items():
            if "running" in k:
                _close(v, g[f"b{i}/after/{k}"], rel=1e-6)
        _close(blk.eval()(x.detach()).detach(), g[f"b{i}/y_eval"], rel=1e-5)


def test_eval_helpers_follow_the_reference_procedures(golden):
    """``evaluate_classification`` (ClassificationModel.py:109-150) and ``evaluate_layercam_on_test_set``
    (LayerCAM.py:84-130) of the PRODUCT against outputs of the reference's own function bodies on stub inputs
    (tests/golden/make_golden.py gen_eval_helpers).  Both are host-side metric procedures around a model / generator
    call, so with stubs they run without a GPU: accuracy and macro-F1; threshold, trimap == 1, nearest resize, the
    11-image cap, IoU / accuracy averaging."""
    from weaklysuperviseddl_amd.TraditionalModel import evaluate_classification, evaluate_layercam_on_test_set
    g = golden("eval_helpers")
    logits, labels = T(g["cls_logits"]), T(g["cls_labels"])

    class Stub(nn.Module):
        k = 0

        def forward(self, x):
            self.k += 1
            return logits[self.k - 1], None

    loader = [(torch.zeros(logits.shape[1], 1), (labels[k], None)) for k in range(logits.shape[0])]
    acc, f1 = evaluate_classification(Stub(), loader, "cpu", num_classes=37, log=None)
    assert abs(acc - g["cls_acc_f1"][0]) < 1e-9 and abs(f1 - g["cls_acc_f1"][1]) < 1e-6

    cams, tri, tri_small, lab = T(g["cam_maps"]), T(g["cam_tri"]).long(), T(g["cam_tri_small"]).long(), T(g["cam_labels"])

    class Gen:
        k = 0

        def generate_batch(self, x, alpha, cls, thresh=None):
            self.k += 1
            cam = cams[self.k - 1:self.k].clone()
            return cam, ((cam >= thresh) & (cam > 0)).to(torch.uint8)       # the fused threshold of wsdl_layercam_epilogue

    n = cams.shape[0]
    test_loader = [(torch.zeros(1, 3, 56, 56), (lab[k:k + 1], (tri_small if k % 3 == 2 else tri)[k])) for k in range(n)]
    gen = Gen()
    res = evaluate_layercam_on_test_set(gen, test_loader, alpha=1.0, cam_thresh=0.3, device="cpu", log=None)
    assert gen.k == 11                                                     # LayerCAM.py:119-120
    assert abs(res["layercam_fg_iou"] - g["cam_iou_acc"][0]) < 1e-9 and abs(res["layercam_fg_acc"] - g["cam_iou_acc"][1]) < 1e-9
