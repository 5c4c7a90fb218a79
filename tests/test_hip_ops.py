"""GPU parity tests of the individual HIP kernels (through the C ABI via ops.py) against the CPU
oracle primitives (torch CPU ATen ops - the arithmetic the reference reaches through torchvision) and the
golden fixtures.  Tolerances: fp32 1e-3 relative (north_star), integer outputs bit-exact."""
import json

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

T = torch.from_numpy


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def rel_err(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def assert_close(a, b, rel=1e-3, what=""):
    assert tuple(a.shape) == tuple(b.shape), (what, a.shape, b.shape)
    e = rel_err(a, b)
    assert e <= rel, f"{what}: rel err {e:.3e} > {rel}"


CONV_CASES = [
    # B, Cin, H, W, Cout, k, stride, pad, dil
    (2, 16, 20, 24, 32, 1, 1, 0, 1),
    (2, 32, 17, 19, 64, 3, 1, 1, 1),
    (2, 32, 16, 16, 48, 3, 1, 2, 2),       # dilated, Cout not a tile multiple
    (1, 64, 12, 12, 32, 3, 1, 12, 12),     # ASPP-like: dilation >= feature size on some taps
    (2, 32, 18, 22, 64, 3, 2, 1, 1),       # stride 2 (layer2.0.conv2)
    (2, 48, 16, 16, 96, 1, 2, 0, 1),       # 1x1 stride 2 (downsample)
    (2, 3, 40, 36, 64, 7, 2, 3, 1),        # stem 7x7 s2, Cin=3 (unaligned K): the stem kernel, partial tiles
    (3, 3, 33, 71, 64, 7, 2, 3, 1),        # ... odd sizes, three column tiles
    (2, 3, 30, 30, 32, 7, 2, 3, 1),        # same geometry, 32 output channels: the generic fp32 kernel
    (3, 64, 9, 11, 2, 1, 1, 0, 1),         # classifier[4]: Cout=2
    (2, 256, 8, 8, 21, 1, 1, 0, 1),        # aux head: Cout=21
    (4, 128, 1, 1, 37, 1, 1, 0, 1),        # fc as 1x1 conv on a 1x1 map
    (2, 160, 14, 14, 192, 3, 1, 1, 1),     # Cout = 128 + 64 (two M tiles, second partial); small grid -> split-K
    (8, 512, 14, 14, 512, 3, 1, 2, 2),     # CAM path layer4 3x3 at B=8: 25 pixel tiles -> split-K x3
    (8, 1024, 14, 14, 2048, 1, 1, 0, 1),   # CAM path layer4.0 downsample
    (3, 64, 32, 32, 128, 3, 1, 12, 12),    # ASPP-like on a 32x32 map: column bands [0,12) [12,20) [20,32)
    (2, 128, 32, 40, 128, 3, 1, 24, 24),   # dilation 24, non-square map: bands [0,16) [16,24) [24,40)
    (2, 128, 32, 32, 256, 3, 1, 1, 1),     # split weight gradient: 9 N tiles, 32-pixel chunks = whole rows
    (3, 128, 9, 11, 256, 3, 1, 2, 2),      # same kernel, chunks straddling rows / images, ragged last chunk
    (2, 256, 16, 16, 128, 3, 2, 1, 1),     # same kernel, stride 2
    (2, 1024, 8, 8, 128, 1, 1, 0, 1),      # same kernel, 1x1 with 8 N tiles
    (2, 256, 9, 7, 1024, 1, 1, 0, 1),      # 1x1 with 2 N tiles but 8 row tiles: role-swapped (dW^T) weight gradient
]


@pytest.fixture(params=["fp16x2", "bf16x3", "fp32"])
def arithmetic(request):
    """Every convolution arithmetic path: the fp16x2-split kernels (default where the shape allows), the bf16x3-split
    kernels (conv_arith = 0) and the fp32-MFMA kernels (wsdl_set_option conv_split / wgrad_split = 0)."""
    from weaklysuperviseddl_amd import ops
    on = int(request.param != "fp32")
    ops.set_option("conv_split", on)
    ops.set_option("wgrad_split", on)
    ops.set_option("conv_arith", int(request.param != "bf16x3"))
    yield request.param
    ops.set_option("conv_split", 1)
    ops.set_option("wgrad_split", 1)
    ops.set_option("conv_arith", 1)


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd_dgrad_wgrad(dev, case, arithmetic):
    from weaklysuperviseddl_amd import ops
    B, Cin, H, W, Cout, k, s, p, d = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    xr, wr = x.clone().requires_grad_(), w.clone().requires_grad_()
    yr = F.conv2d(xr, wr, None, s, p, d)
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy)

    xd, wd_, dyd = x.to(dev), w.to(dev), dy.to(dev)
    wf, wdg = ops.prep_weights(wd_)
    y = ops.conv2d_fwd(xd, wf, w.shape, s, p, d)
    assert_close(y, yr, what="fwd")
    dx = ops.conv2d_dgrad(dyd, wdg, w.shape, x.shape, s, p, d)
    assert_close(dx, xr.grad, what="dgrad")
    dw = ops.conv2d_wgrad(xd, dyd, w.shape, s, p, d)
    assert_close(dw, wr.grad, what="wgrad")
    # accumulate forms
    dx2 = ops.conv2d_dgrad(dyd, wdg, w.shape, x.shape, s, p, d, accumulate_into=dx.clone())
    assert_close(dx2, 2 * xr.grad, what="dgrad accumulate")
    if x.numel() % 8 == 0:
        # masked accumulate (the identity branch of a bottleneck): dx = dgrad + [bit] * previous, exactly
        prev = torch.randn(x.shape, generator=g)
        keep = torch.rand(x.shape, generator=g) < 0.5
        bits = T(np.packbits(keep.numpy().reshape(-1), bitorder="little")).to(dev)
        dx3 = ops.conv2d_dgrad(dyd, wdg, w.shape, x.shape, s, p, d, accumulate_into=prev.to(dev), acc_mask=bits)
        dx4 = ops.conv2d_dgrad(dyd, wdg, w.shape, x.shape, s, p, d, accumulate_into=(prev * keep).to(dev))
        assert torch.equal(dx3, dx4), "masked accumulate"
        with pytest.raises(ops.WsdlError):
            ops.conv2d_dgrad(dyd, wdg, w.shape, x.shape, s, p, d, acc_mask=bits)
    dw2 = ops.conv2d_wgrad(xd, dyd, w.shape, s, p, d, out=dw.clone(), accumulate=True)
    assert_close(dw2, 2 * wr.grad, what="wgrad accumulate")


MODES = {"fp32": dict(conv_split=0, wgrad_split=0, conv_arith=1),
         "bf16x3": dict(conv_split=1, wgrad_split=1, conv_arith=0),
         # fp16x2 with the low piece carried at 2^11 and the cross products in their own accumulator (conv_split.h: "AR = 2"):
         # forward / input gradient only (the weight-gradient kernels keep the plain fp16x2 arithmetic)
         "fp16x2s": dict(conv_split=1, wgrad_split=1, conv_arith=2),
         "fp16x2": dict(conv_split=1, wgrad_split=1, conv_arith=1)}     # the default


@pytest.mark.parametrize("data", ["unit", "wide"])
def test_split_arithmetic_is_fp32_accurate(dev, data):
    """The split kernels claim fp32-level accuracy: against a float64 convolution their rms error must not exceed the
    fp32-MFMA kernels' (exact fp32 fma chains) by more than a small factor, in every pass - on unit-variance data and
    on "wide" data (channels spanning seven decades, gradients around 1e-6) that exercises the fp16x2 kernels' per-tensor
    power-of-two scales.  Measured (tools/conv_accuracy.py): fp16x2 0.6-2.0 x the fp32 chain's error, never above
    1.2e-7 relative - one fp32 rounding of the output."""
    from weaklysuperviseddl_amd import ops
    g = torch.Generator().manual_seed(77)
    errs = {}
    try:
        for Cin, Cout, k, s, d, H, B in [(256, 256, 3, 1, 2, 32, 2), (1024, 256, 1, 1, 1, 16, 4), (128, 128, 3, 2, 1, 32, 2),
                                         (256, 1024, 1, 1, 1, 16, 4)]:
            pad = (k // 2) * d if k > 1 else 0
            x = torch.randn(B, Cin, H, H, generator=g).to(dev)
            w = (torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5).to(dev)
            if data == "wide":
                x = torch.relu(x) * torch.logspace(-4, 3, Cin, device=dev).view(1, Cin, 1, 1)
                w = w * torch.logspace(-2, 2, Cout, device=dev).view(Cout, 1, 1, 1)
            ref = F.conv2d(x.double(), w.double(), None, s, pad, d)
            dy = torch.randn(ref.shape, generator=g).to(dev)
            if data == "wide":
                dy = dy * 1e-6 * torch.logspace(-3, 3, dy.shape[-1], device=dev)
            ref_dx = torch.nn.grad.conv2d_input(x.shape, w.double(), dy.double(), s, pad, d)
            ref_dw = torch.nn.grad.conv2d_weight(x.double(), w.shape, dy.double(), s, pad, d)
            for mode, opts in MODES.items():
                for o, v in opts.items():
                    ops.set_option(o, v)
                wf, wdg = ops.prep_weights(w)
                outs = (ops.conv2d_fwd(x, wf, w.shape, s, pad, d), ops.conv2d_dgrad(dy, wdg, w.shape, x.shape, s, pad, d),
                        ops.conv2d_wgrad(x, dy, w.shape, s, pad, d))
                for name, o, r in zip(("fwd", "dgrad", "wgrad"), outs, (ref, ref_dx, ref_dw)):
                    rms = ((o.double() - r).pow(2).mean().sqrt() / r.pow(2).mean().sqrt()).item()
                    errs[(Cin, k, name, mode)] = rms
    finally:
        for o, v in MODES["fp16x2"].items():
            ops.set_option(o, v)
    for (Cin, k, name, mode), e in errs.items():
        assert e < 5e-6, (Cin, k, name, mode, e)
        if mode != "fp32":
            assert e < 3.0 * errs[(Cin, k, name, "fp32")] + 1e-8, (Cin, k, name, mode, e, errs[(Cin, k, name, "fp32")])


def _region_maxnorm_ratio(o, r, blk=8):
    """Worst region of an output tensor (B,C,H,W): max |o - r| over the region / max |r| over the SAME region, for
    (i) every output channel and (ii) every blk x blk spatial block (all channels) - a tensor-wide rms cannot see a
    channel or a corner whose values are all small."""
    e, a = (o.double() - r).abs(), r.abs()
    per_ch = (e.amax(dim=(0, 2, 3)) / (a.amax(dim=(0, 2, 3)) + 1e-300)).max().item()
    B, C, H, W = e.shape
    Hb, Wb = H // blk * blk, W // blk * blk
    eb = e[:, :, :Hb, :Wb].reshape(B, C, Hb // blk, blk, Wb // blk, blk).amax(dim=(1, 3, 5))
    ab = a[:, :, :Hb, :Wb].reshape(B, C, Hb // blk, blk, Wb // blk, blk).amax(dim=(1, 3, 5))
    return per_ch, (eb / (ab + 1e-300)).max().item()


@pytest.mark.parametrize("data", ["unit", "outlier20", "outlier30", "outlier35", "graded30", "wide"])
def test_split_arithmetic_max_norm_per_channel_and_per_block(dev, data):
    """The fp16x2 kernels scale each operand TENSOR by one power of two (conv_split.h): elements more than 2^17 below
    the tensor's maximum keep fewer than 22 bits (they stay exact to 2^-39 of the maximum).  A tensor-wide rms hides
    that; this test takes the max-norm error of every output channel and of every 8 x 8 spatial block against float64,
    relative to that region's own maximum:
      unit      - unit-variance operands: every region at fp32 level (<= 4 x the exact-fp32 MFMA kernels' figure);
      outlierE  - ONE activation element of 2^E among unit ones (image 0, far corner): regions that never see the
                  outlier read operands 2^E below the tensor maximum;
      graded30  - images whose magnitudes fall by 2^30 across the batch (a gradient tensor with confident and unconfident images);
      wide      - channel magnitudes spanning seven decades (the rms test's data): as unit (the small channels contribute
                  nothing visible to any output region, the large ones keep their 22 bits).
    Two arithmetics are judged:
      fp16x2  (conv_arith = 1, the default): the documented floor - a region 2^E below the maximum is computed to 2^-(38-E)
              of its own maximum (outlier20: <= 2^-17, two orders inside north_star's 1e-3; outlier30 / graded30: ~1e-3, past it);
      fp16x2s (conv_arith = 2, the guard: the low piece at 2^11 with its own accumulator - same three MFMAs, 64 more
              registers, 6.6 % on the training step, profiles/r04_notes.md): EVERY region of every case <= 1e-4 - measured
              3e-7 (unit, better than fp16x2: the cross products no longer round into the large accumulator), 9e-7
              (outlier30), 2.6e-5 (outlier35), 2.8e-6 (graded30); its own floor is 2^-50 of the maximum."""
    from weaklysuperviseddl_amd import ops
    g = torch.Generator().manual_seed(78)
    worst = {}
    E = int(data[7:]) if data.startswith("outlier") else 0
    try:
        for Cin, Cout, k, s, d, H, B in [(256, 256, 3, 1, 2, 32, 4), (1024, 256, 1, 1, 1, 16, 4)]:
            pad = (k // 2) * d if k > 1 else 0
            x = torch.randn(B, Cin, H, H, generator=g).to(dev)
            w = (torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5).to(dev)
            if E:
                x[0, 3, H - 1, H - 1] = 2.0 ** E
            if data == "wide":
                x = torch.relu(x) * torch.logspace(-4, 3, Cin, device=dev).view(1, Cin, 1, 1)
                w = w * torch.logspace(-2, 2, Cout, device=dev).view(Cout, 1, 1, 1)
            grade = torch.tensor([2.0 ** (-30.0 * b / (B - 1)) for b in range(B)], device=dev).view(B, 1, 1, 1)
            if data == "graded30":
                x = x * grade
            ref = F.conv2d(x.double(), w.double(), None, s, pad, d)
            dy = torch.randn(ref.shape, generator=g).to(dev)
            if E:
                dy[0, 5, 0, 0] = 2.0 ** E
            if data == "graded30":
                dy = dy * grade
            ref_dx = torch.nn.grad.conv2d_input(x.shape, w.double(), dy.double(), s, pad, d)
            for mode in ("fp32", "fp16x2", "fp16x2s"):
                for o, v in MODES[mode].items():
                    ops.set_option(o, v)
                wf, wdg = ops.prep_weights(w)
                y = ops.conv2d_fwd(x, wf, w.shape, s, pad, d)
                dx = ops.conv2d_dgrad(dy, wdg, w.shape, x.shape, s, pad, d)
                for name, o, r in (("fwd", y, ref), ("dgrad", dx, ref_dx)):
                    worst[(Cin, k, name, mode)] = _region_maxnorm_ratio(o, r)
    finally:
        for o, v in MODES["fp16x2"].items():
            ops.set_option(o, v)
    for (Cin, k, name, mode), (ch, blk) in sorted(worst.items()):
        print("%-10s Cin %4d k %d %-5s %-10s worst channel %.2e  worst 8x8 block %.2e" % (data, Cin, k, name, mode, ch, blk))
    from conftest import report_line
    report_line("split arithmetic, %-9s worst 8x8 block vs float64: " % data + ", ".join(
        "%s %.1e" % (m, max(v[1] for kk, v in worst.items() if kk[3] == m)) for m in ("fp32", "fp16x2", "fp16x2s")))
    for (Cin, k, name, mode), (ch, blk) in worst.items():
        f_ch, f_blk = worst[(Cin, k, name, "fp32")]
        if mode == "fp16x2":
            # measured (GPUTEST r3 / r4): unit / wide 0.5-0.9 x the fp32 kernels' figure in every region; outlier20 worst block
            # 1.2e-6 - 1.5e-6 against 5e-7 - 8e-7 (the 2^-19 floor of operands 2^20 below the tensor maximum, averaged over K);
            # outlier30 1.3e-3, outlier35 4.8e-2, graded30 5e-3: the documented floor 2^-(39-E), past north_star's 1e-3 from E ~ 29
            if E or data == "graded30":
                bound_ch = bound_blk = 2.0 ** -(37 - (E or 30))
            else:
                bound_ch, bound_blk = 4 * f_ch + 1e-7, 4 * f_blk + 1e-7
            assert ch <= bound_ch and blk <= bound_blk, (data, Cin, k, name, ch, blk, f_ch, f_blk)
        elif mode.startswith("fp16x2s"):
            bound = 1e-4 if (E >= 30 or data == "graded30") else 4 * max(f_ch, f_blk) + 1e-7
            assert ch <= bound and blk <= bound, (data, Cin, k, name, mode, ch, blk, f_ch, f_blk)


def test_fp16x2_scales_cover_extreme_magnitudes(dev):
    """The per-tensor scale is a power of two taken from the tensor's amax: results must not depend on the absolute
    magnitude of either operand (fp16 alone would overflow above 65504 and flush below 6e-8)."""
    from weaklysuperviseddl_amd import ops
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 128, 16, 16, generator=g).to(dev)
    w = (torch.randn(128, 128, 3, 3, generator=g) * 0.03).to(dev)
    dy = torch.randn(2, 128, 16, 16, generator=g).to(dev)
    wf, wd = ops.prep_weights(w)
    base = (ops.conv2d_fwd(x, wf, w.shape, 1, 1, 1), ops.conv2d_dgrad(dy, wd, w.shape, x.shape, 1, 1, 1),
            ops.conv2d_wgrad(x, dy, w.shape, 1, 1, 1))
    for sx, sw in ((2.0 ** 40, 2.0 ** -30), (2.0 ** -40, 2.0 ** 20), (2.0 ** 60, 2.0 ** -55)):
        xs, ws, dys = x * sx, w * sw, dy * sx
        wf2, wd2 = ops.prep_weights(ws)
        got = (ops.conv2d_fwd(xs, wf2, w.shape, 1, 1, 1), ops.conv2d_dgrad(dys, wd2, w.shape, x.shape, 1, 1, 1),
               ops.conv2d_wgrad(xs, dys, w.shape, 1, 1, 1))
        for o, b, f in zip(got, base, (sx * sw, sx * sw, sx * sx)):
            assert torch.equal(o, b * f), (sx, sw)          # power-of-two scalings commute with every rounding involved
    z = ops.conv2d_fwd(torch.zeros_like(x), wf, w.shape, 1, 1, 1)           # amax = 0: scale 1, exact zeros
    assert z.abs().max().item() == 0.0
    bad = x.clone()
    bad[0, 0, 0, 0] = float("inf")
    assert not torch.isfinite(ops.conv2d_fwd(bad, wf, w.shape, 1, 1, 1)).all()   # an inf input is not silently dropped


@pytest.mark.parametrize("opt,val", [("wgrad_xcd", 1), ("wgrad_xcd", 2), ("xcd_map", 0)])
def test_scheduling_options_reproduce_the_default_bit_for_bit(dev, opt, val):
    """Options that change how operands travel (XCD-aware tile orders) and not what is computed: forward, input gradient and weight gradient equal the default's bit for bit, on shapes that
    reach the 256x128 and the 4-wave forms, a dilated 3x3 with masked row ends, a strided conv and a ragged one."""
    from weaklysuperviseddl_amd import ops
    shapes = [(16, 512, 512, 3, 1, 2, 32), (4, 256, 256, 3, 1, 4, 16), (2, 128, 128, 3, 2, 1, 32), (16, 256, 1024, 1, 1, 1, 32),
              (3, 48, 80, 3, 1, 1, 17), (2, 128, 256, 3, 1, 36, 24)]
    default = {"wgrad_xcd": 0, "xcd_map": 1}[opt]          # (wgrad_xcd: compared against the launch order)
    try:
        for B, Cin, Cout, k, st, d, H in shapes:
            g = torch.Generator(device=dev).manual_seed(Cin + Cout + H)
            pad = (k // 2) * d
            x = torch.randn(B, Cin, H, H, device=dev, generator=g)
            w = torch.randn(Cout, Cin, k, k, device=dev, generator=g) * 0.05
            res = []
            for v in (default, val):
                ops.set_option(opt, v)
                wf, wd = ops.prep_weights(w)
                y = ops.conv2d_fwd(x, wf, w.shape, st, pad, d)
                dy = torch.randn(y.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
                res.append((y, ops.conv2d_dgrad(dy, wd, w.shape, x.shape, st, pad, d), ops.conv2d_wgrad(x, dy, w.shape, st, pad, d)))
            for a, b, what in zip(res[0], res[1], ("fwd", "dgrad", "wgrad")):
                assert torch.equal(a, b), (opt, val, what, (B, Cin, Cout, k, st, d, H))
    finally:
        ops.set_option(opt, default)


@pytest.mark.parametrize("opts", [dict(wgrad_min_tiles=6)])
def test_other_mfma_shapes_and_tiles_agree_with_the_default(dev, opts):
    """Options that change which shapes the fp16x2 weight-gradient kernel takes (from six 128-wide N tiles on - the default of
    rounds 2-5 - instead of from one: another summation order): results within a few fp32 roundings of the default's, pass by pass (each is measured against float64
    in test_split_arithmetic_is_fp32_accurate).  (The 32x32x16 forms of the K-chunk-32 kernels - conv_mfma16 / wgrad_mfma16 = 0
    - were compared here until round 4 removed them.)"""
    from weaklysuperviseddl_amd import ops
    defaults = dict(wgrad_min_tiles=1)
    shapes = [(16, 512, 512, 3, 1, 2, 32), (4, 64, 64, 3, 1, 1, 32), (4, 64, 256, 1, 1, 1, 32), (4, 256, 64, 1, 1, 1, 32),
              (8, 256, 128, 1, 1, 1, 32), (2, 128, 128, 3, 2, 1, 32), (3, 192, 320, 3, 1, 2, 24)]
    try:
        for B, Cin, Cout, k, st, d, H in shapes:
            g = torch.Generator(device=dev).manual_seed(Cin + Cout + H)
            pad = (k // 2) * d
            x = torch.randn(B, Cin, H, H, device=dev, generator=g)
            w = torch.randn(Cout, Cin, k, k, device=dev, generator=g) * 0.05
            res = []
            for cfg in (defaults, {**defaults, **opts}):
                for o, v in cfg.items():
                    ops.set_option(o, v)
                wf, wd = ops.prep_weights(w)
                y = ops.conv2d_fwd(x, wf, w.shape, st, pad, d)
                dy = torch.randn(y.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
                res.append((y, ops.conv2d_dgrad(dy, wd, w.shape, x.shape, st, pad, d), ops.conv2d_wgrad(x, dy, w.shape, st, pad, d)))
            for a, b, what in zip(res[0], res[1], ("fwd", "dgrad", "wgrad")):
                err = ((a - b).abs().max() / a.abs().max()).item()
                assert err < 2e-6, (opts, what, (B, Cin, Cout, k, st, d, H), err)
    finally:
        for o, v in defaults.items():
            ops.set_option(o, v)


def test_weight_layout_sizes(dev):
    """The opaque layout buffers: 4 bytes per weight (fp32 k-major; two fp16 pieces + a 16-byte amax trailer) or 6 (three
    bf16 pieces + trailer), by shape and option."""
    import ctypes
    from weaklysuperviseddl_amd import ops
    from weaklysuperviseddl_amd._lib import lib
    plain = ctypes.c_int(-1)
    n = 64 * 32 * 9
    assert lib().wsdl_conv2d_weight_layout_bytes(64, 32, 3, 3, 0, ctypes.byref(plain)) == n * 4 + 16 and plain.value == 0
    ops.set_option("conv_arith", 0)
    assert lib().wsdl_conv2d_weight_layout_bytes(64, 32, 3, 3, 0, ctypes.byref(plain)) == n * 6 + 16 and plain.value == 0
    ops.set_option("conv_arith", 1)
    assert lib().wsdl_conv2d_weight_layout_bytes(64, 3, 7, 7, 0, ctypes.byref(plain)) == 64 * 3 * 49 * 4 and plain.value == 1
    assert lib().wsdl_conv2d_weight_layout_bytes(2, 256, 1, 1, 1, ctypes.byref(plain)) == 2 * 256 * 4 and plain.value == 1
    ops.set_option("conv_split", 0)
    try:
        assert lib().wsdl_conv2d_weight_layout_bytes(64, 32, 3, 3, 0, ctypes.byref(plain)) == n * 4 and plain.value == 1
    finally:
        ops.set_option("conv_split", 1)


def test_conv_epilogue_scale_shift_residual_relu(dev):
    from weaklysuperviseddl_amd import ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 32, 10, 10, generator=g)
    w = torch.randn(64, 32, 3, 3, generator=g) * 0.1
    sc, sh = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g)
    res = torch.randn(2, 64, 10, 10, generator=g)
    ref = F.relu(F.conv2d(x, w, None, 1, 1, 1) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1) + res)
    wf, _ = ops.prep_weights(w.to(dev), True, False)
    y = ops.conv2d_fwd(x.to(dev), wf, w.shape, 1, 1, 1, sc.to(dev), sh.to(dev), res.to(dev), True)
    assert_close(y, ref, what="fused epilogue")
    # the same epilogue through the split-K path (tiny grid: partial sums in slabs, epilogue in the reduce kernel)
    x2 = torch.randn(2, 64, 12, 12, generator=g)
    w2 = torch.randn(128, 64, 3, 3, generator=g) * 0.1
    sc2, sh2, res2 = torch.rand(128, generator=g) + 0.5, torch.randn(128, generator=g), torch.randn(2, 128, 12, 12, generator=g)
    ref2 = F.relu(F.conv2d(x2, w2, None, 1, 1, 1) * sc2.view(1, -1, 1, 1) + sh2.view(1, -1, 1, 1) + res2)
    from weaklysuperviseddl_amd._lib import lib
    assert lib().wsdl_conv2d_igemm_workspace(2, 64, 12, 12, 128, 3, 3, 1, 1, 1, 0) > 0
    wf2, _ = ops.prep_weights(w2.to(dev), True, False)
    y2 = ops.conv2d_fwd(x2.to(dev), wf2, w2.shape, 1, 1, 1, sc2.to(dev), sh2.to(dev), res2.to(dev), True)
    assert_close(y2, ref2, what="fused epilogue, split-K")
    # strided output (channel slice of a wider tensor)
    wide = torch.zeros(2, 100, 10, 10, device=dev)
    ops.conv2d_fwd(x.to(dev), wf, w.shape, 1, 1, 1, out=wide[:, 20:84])
    assert_close(wide[:, 20:84], F.conv2d(x, w, None, 1, 1, 1), what="sliced output")
    assert wide[:, :20].abs().max().item() == 0 and wide[:, 84:].abs().max().item() == 0


def test_conv_bad_geometry_raises(dev):
    from weaklysuperviseddl_amd import ops, WsdlError
    x = torch.randn(1, 8, 4, 4, device=dev)
    w = torch.randn(8, 8, 3, 3, device=dev)
    wf, _ = ops.prep_weights(w)
    with pytest.raises(WsdlError):
        ops.conv2d_fwd(x, wf, (8, 4, 3, 3), 1, 1, 1)          # channel mismatch
    with pytest.raises(WsdlError):
        ops.conv2d_fwd(x, wf, w.shape, 1, 0, 4)               # empty output
    with pytest.raises(WsdlError):
        ops.conv2d_fwd(x.cpu(), wf, w.shape, 1, 1, 1)         # host tensor: no CPU fallback


@pytest.mark.parametrize("shape,relu,res", [((4, 24, 9, 7), True, True), ((3, 64, 16, 16), True, False),
                                            ((8, 40, 1, 1), False, False), ((2, 16, 12, 20), False, True),
                                            # channel-resident fused kernels (C >= 192, B*HW <= 32768, HW % 4 == 0)
                                            ((4, 256, 8, 8), True, True), ((2, 192, 16, 16), False, False),
                                            ((3, 200, 12, 4), True, False), ((24, 192, 32, 32), True, True),
                                            # the ReLU mask as bits: two-pass form with odd run lengths, the three resident forms
                                            ((3, 40, 8, 8), True, True), ((7, 16, 16, 24), True, True),
                                            ((5, 256, 4, 6), True, True), ((16, 320, 32, 32), True, True),
                                            ((16, 1024, 16, 16), True, True),
                                            # 64 / 128 channels in the resident forms (forward 1024 x 16 float4 at 16 x 64 x 64)
                                            ((16, 64, 64, 64), True, False), ((16, 128, 32, 32), True, True),
                                            ((16, 256, 64, 64), True, True)])
def test_batchnorm_train_fwd_bwd(dev, shape, relu, res):
    from weaklysuperviseddl_amd import ops
    g = torch.Generator().manual_seed(11)
    B, C, H, W = shape
    x = (torch.randn(shape, generator=g) * 2 + 0.7)
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    rm, rv = torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5
    r = torch.randn(shape, generator=g) if res else None
    xr, gr, br = x.clone().requires_grad_(), gamma.clone().requires_grad_(), beta.clone().requires_grad_()
    rr = r.clone().requires_grad_() if res else None
    rm_ref, rv_ref = rm.clone(), rv.clone()
    yr = F.batch_norm(xr, rm_ref, rv_ref, gr, br, True, 0.1, 1e-5)
    if res:
        yr = yr + rr
    pre = yr
    dy = torch.randn(shape, generator=g)
    rm_d, rv_d = rm.to(dev), rv.to(dev)
    y, mean, invstd = ops.bn_train_fwd(x.to(dev), gamma.to(dev), beta.to(dev), rm_d, rv_d, 0.1, 1e-5,
                                       r.to(dev) if res else None, relu)
    if relu:
        # An element whose pre-activation is within rounding of zero may take the other side of the ReLU in another fp32
        # implementation (a handful among the 16.8 million of the largest case) - and one such element moves its channel's
        # dgamma by more than 1e-3.  The reference backward therefore runs through OUR mask, after checking that the two
        # masks differ only there.
        ours, theirs = (y.cpu() > 0), (pre.detach() > 0)
        differ = ours != theirs
        assert (pre.detach().abs()[differ] <= 1e-5).all() and differ.float().mean().item() < 1e-5, int(differ.sum())
        yr = pre * ours
    safe = torch.ones_like(pre.detach(), dtype=torch.bool)
    yr.backward(dy)
    assert_close(y, yr, what="bn fwd")
    assert_close(rm_d, rm_ref, what="running mean")
    assert_close(rv_d, rv_ref, what="running var")
    dx, dgamma, dbeta, dres = ops.bn_train_bwd(x.to(dev), dy.to(dev), y, gamma.to(dev), mean, invstd, relu, res)
    assert_close(dx.cpu() * safe, xr.grad * safe, what="bn dx")
    assert_close(dgamma, gr.grad, what="bn dgamma")
    assert_close(dbeta, br.grad, what="bn dbeta")
    if res:
        assert_close(dres.cpu() * safe, rr.grad * safe, what="bn dres")
    if relu and res:
        # the mask as bits from the forward kernel (what the fused conv -> BN -> +residual node keeps instead of y)
        rm2, rv2 = rm.to(dev), rv.to(dev)
        y2, mean2, invstd2, bits = ops.bn_train_fwd(x.to(dev), gamma.to(dev), beta.to(dev), rm2, rv2, 0.1, 1e-5, r.to(dev), True,
                                                    want_mask=True)
        assert torch.equal(y2, y) and torch.equal(mean2, mean) and torch.equal(invstd2, invstd)
        if (H * W) % 8 == 0:
            want = np.packbits((y.cpu().numpy() > 0).reshape(-1), bitorder="little")
            assert bits is not None and bits.dtype == torch.uint8 and np.array_equal(bits.cpu().numpy(), want)
            dx3, dg3, db3, dres3 = ops.bn_train_bwd(x.to(dev), dy.to(dev), None, gamma.to(dev), mean, invstd, True, True,
                                                    relu_mask=bits)
            assert torch.equal(dx3, dx) and torch.equal(dg3, dgamma) and torch.equal(db3, dbeta) and torch.equal(dres3, dres)
            with pytest.raises(ops.WsdlError):
                ops.bn_train_bwd(x.to(dev), dy.to(dev), None, gamma.to(dev), mean, invstd, True, True, relu_mask=bits[:-1])
        else:
            assert bits is None
    if relu and not res:
        # the mask recomputed from x (what the fused conv -> BN node uses: y is neither kept nor read) is THE SAME mask
        dx2, dg2, db2, _ = ops.bn_train_bwd(x.to(dev), dy.to(dev), None, gamma.to(dev), mean, invstd, True, False,
                                            beta=beta.to(dev))
        assert torch.equal(dx2, dx) and torch.equal(dg2, dgamma) and torch.equal(db2, dbeta)
        assert torch.equal((y > 0), (torch.nn.functional.relu(y) > 0))


def test_bn_fold_and_affine_bwd(dev):
    from weaklysuperviseddl_amd import ops
    g = torch.Generator().manual_seed(5)
    C = 20
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    rm, rv = torch.randn(C, generator=g), torch.rand(C, generator=g) + 0.2
    sc, sh = ops.bn_fold(gamma.to(dev), beta.to(dev), rm.to(dev), rv.to(dev), 1e-5)
    assert_close(sc, gamma / torch.sqrt(rv + 1e-5), rel=1e-6)
    assert_close(sh, beta - rm * gamma / torch.sqrt(rv + 1e-5), rel=1e-5)
    for hw in ((5, 5), (14, 14), (6, 10), (64, 48)):         # plane-per-workgroup form; flat float4 form (HW % 4 == 0, small); large planes
        y = torch.randn(3, C, *hw, generator=g)
        dy = torch.randn(3, C, *hw, generator=g)
        dconv, dres = ops.affine_act_bwd(dy.to(dev), y.to(dev), sc, True, True, True)
        m = (y > 0).float()
        assert_close(dres, dy * m, rel=1e-6)
        assert_close(dconv, dy * m * sc.cpu().view(1, -1, 1, 1), rel=1e-6)
        assert abs(dconv._wsdl_amax.item() - dconv.abs().max().item()) <= 1e-6 * dconv.abs().max().item()
        d2, _ = ops.affine_act_bwd(dy.to(dev), y.to(dev), None, True, True, False)      # plain ReLU backward
        assert_close(d2, dy * m, rel=1e-6)


@pytest.mark.parametrize("hw", [(16, 16), (15, 17), (7, 7), (112, 112)])
def test_maxpool_fwd_bwd(dev, hw):
    from weaklysuperviseddl_amd import ops
    g = torch.Generator().manual_seed(hw[0])
    x = F.relu(torch.randn(2, 5, *hw, generator=g))     # post-ReLU input: many ties at zero
    xr = x.clone().requires_grad_()
    yr = F.max_pool2d(xr, 3, 2, 1)
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy)
    xd = x.to(dev).requires_grad_()
    y = ops.max_pool_3x3_s2(xd)
    assert torch.equal(y.cpu(), yr.detach())
    y.backward(dy.to(dev))
    # ties at zero may route gradient to a different zero element; everything else must agree
    nz = x > 0
    assert_close(xd.grad.cpu() * nz, xr.grad * nz, rel=1e-6, what="maxpool dx")
    assert abs(xd.grad.sum().item() - xr.grad.sum().item()) < 1e-3


def test_global_avgpool_and_linear(dev):
    from weaklysuperviseddl_amd import ops
    g = torch.Generator().manual_seed(2)
    x = torch.randn(3, 40, 7, 9, generator=g)
    w, b = torch.randn(11, 40, generator=g) * 0.2, torch.randn(11, generator=g)
    xr, wr, br = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    out_r = F.linear(F.adaptive_avg_pool2d(xr, 1).flatten(1), wr, br)
    dy = torch.randn(3, 11, generator=g)
    out_r.backward(dy)
    xd, wd_, bd = x.to(dev).requires_grad_(), w.to(dev).requires_grad_(), b.to(dev).requires_grad_()
    out = ops.linear(ops.global_avg_pool(xd).flatten(1), wd_, bd)
    assert_close(out, out_r, what="linear fwd")
    out.backward(dy.to(dev))
    assert_close(xd.grad, xr.grad, what="dx")
    assert_close(wd_.grad, wr.grad, what="dw")
    assert_close(bd.grad, br.grad, what="db")


def test_global_avgpool_one_wave_per_plane(dev):
    """ASPP's pooling branch (2048 planes of 32 x 32 per image): gap_fwd_wave_kernel - one wave per plane, 16-byte loads - against
    torch's float64 mean, and bitwise reproducible."""
    from weaklysuperviseddl_amd import ops
    x = torch.randn(2, 2048, 32, 32, generator=torch.Generator().manual_seed(3)) + 0.25
    xd = x.to(dev)
    y = ops.global_avg_pool(xd)
    ref = x.double().mean(dim=(2, 3), keepdim=True)
    assert tuple(y.shape) == (2, 2048, 1, 1)
    assert ((y.cpu().double() - ref).abs().max() / ref.abs().max()).item() < 1e-6
    assert torch.equal(y, ops.global_avg_pool(xd))


@pytest.mark.parametrize("shape,size", [((2, 3, 4, 4), (32, 32)), ((1, 2, 14, 14), (224, 224)),
                                        ((2, 5, 1, 1), (8, 8)), ((1, 2, 7, 9), (20, 31)), ((2, 2, 8, 8), (8, 8)),
                                        ((3, 4, 1, 1), (32, 32))])
def test_bilinear_fwd_bwd(dev, shape, size):
    from weaklysuperviseddl_amd import ops
    g = torch.Generator().manual_seed(shape[2])
    x = torch.randn(shape, generator=g)
    xr = x.clone().requires_grad_()
    yr = F.interpolate(xr, size=size, mode="bilinear", align_corners=False)
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy)
    xd = x.to(dev).requires_grad_()
    y = ops.bilinear_resize(xd, size)
    assert_close(y, yr, rel=1e-5, what="bilinear fwd")
    y.backward(dy.to(dev))
    assert_close(xd.grad, xr.grad, rel=1e-4, what="bilinear bwd")


def test_dropout_injected_mask_and_rng(dev):
    from weaklysuperviseddl_amd import ops
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 8, 6, 6, generator=g)
    mask = (torch.rand(x.shape, generator=g) > 0.5).to(torch.uint8)
    xd = x.to(dev).requires_grad_()
    y = ops.dropout(xd, 0.5, True, mask=mask.to(dev))
    assert_close(y, x * mask * 2.0, rel=1e-6)
    dy = torch.randn(x.shape, generator=g)
    y.backward(dy.to(dev))
    assert_close(xd.grad, dy * mask * 2.0, rel=1e-6)
    big = torch.ones(1 << 20, device=dev).view(1, 1, 1024, 1024)
    z = ops.dropout(big, 0.5, True, seed=1234)
    keep = (z > 0).float().mean().item()
    assert abs(keep - 0.5) < 5e-3 and abs(z.max().item() - 2.0) < 1e-6
    z2 = ops.dropout(big, 0.5, True, seed=1234)
    z3 = ops.dropout(big, 0.5, True, seed=1235)
    assert torch.equal(z, z2) and not torch.equal(z, z3)
    assert ops.dropout(big, 0.5, False) is big                    # eval: identity


def test_concat_and_add(dev):
    from weaklysuperviseddl_amd import ops
    g = torch.Generator().manual_seed(4)
    xs = [torch.randn(2, c, 5, 6, generator=g) for c in (3, 8, 1)]
    xd = [t.to(dev).requires_grad_() for t in xs]
    y = ops.concat_channels(xd)
    assert torch.equal(y.cpu(), torch.cat(xs, 1))
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy.to(dev))
    off = 0
    for t, c in zip(xd, (3, 8, 1)):
        assert torch.equal(t.grad.cpu(), dy[:, off:off + c])
        off += c
    a, b = xs[1], torch.randn(2, 8, 5, 6, generator=g)
    ad, bd = a.to(dev).requires_grad_(), b.to(dev).requires_grad_()
    z = ops.add_act(ad, bd, True)
    assert_close(z, F.relu(a + b), rel=1e-6)
    z.backward(torch.ones_like(z))
    assert_close(ad.grad, ((a + b) > 0).float(), rel=1e-6)


@pytest.mark.parametrize("shape", [(2, 2, 16, 16), (3, 5, 9, 13), (1, 21, 8, 8)])
def test_cross_entropy(dev, shape):
    from weaklysuperviseddl_amd import ops
    g = torch.Generator().manual_seed(shape[1])
    B, C, H, W = shape
    logits = torch.randn(shape, generator=g) * 3
    labels = torch.randint(0, C, (B, H, W), generator=g)
    lr = logits.clone().requires_grad_()
    ref = F.cross_entropy(lr, labels)
    (ref * 0.7).backward()
    ld = logits.to(dev).requires_grad_()
    loss = ops.cross_entropy(ld, labels.to(dev))
    assert_close(loss, ref.detach(), rel=1e-5, what="ce loss")
    (loss * 0.7).backward()
    assert_close(ld.grad, lr.grad, rel=1e-4, what="ce grad")


def test_pairwise_loss_vs_golden(dev, golden):
    from weaklysuperviseddl_amd.TraditionalModel import LocalNormalizedCutLoss, ConstrainToBoundaryLossSingle
    g = golden("losses")
    for m in json.loads(str(g["meta"])):
        i = m["idx"]
        if m["kind"] == "ncut":
            preds = T(g[f"ncut{i}_preds"]).to(dev).requires_grad_()
            loss = LocalNormalizedCutLoss(m["sigma_color"], m["window"])(preds, T(g[f"ncut{i}_image"]).to(dev))
            loss.backward()
            assert_close(loss, T(g[f"ncut{i}_loss"]), what=f"ncut{i} loss")
            assert_close(preds.grad, T(g[f"ncut{i}_grad"]), what=f"ncut{i} grad")
        else:
            p = T(g[f"bnd{i}_preds"]).to(dev).requires_grad_()
            loss = ConstrainToBoundaryLossSingle(m["sigma_color"], m["sigma_space"], m["window"])(
                p, T(g[f"bnd{i}_image"]).to(dev))
            loss.backward()
            assert_close(loss, T(g[f"bnd{i}_loss"]), what=f"bnd{i} loss")
            assert_close(p.grad, T(g[f"bnd{i}_grad"]), what=f"bnd{i} grad")
    # 3-D path
    preds = T(g["ncut3d_preds"]).to(dev).requires_grad_()
    loss = LocalNormalizedCutLoss(0.1, 5)(preds, T(g["ncut3d_image"]).to(dev))
    loss.backward()
    assert_close(loss, T(g["ncut3d_loss"]), what="ncut3d loss")
    assert_close(preds.grad, T(g["ncut3d_grad"]), what="ncut3d grad")


def test_compute_affinities_vs_golden(dev, golden):
    from weaklysuperviseddl_amd.TraditionalModel import compute_affinities, ConstrainToBoundaryLossSingle
    g = golden("losses")
    aff = compute_affinities(T(g["aff_image"]).to(dev), 0.1, 5, 5)
    assert len(aff) == 24 and tuple(aff[0].shape) == (2, 1, 16, 16)
    assert_close(torch.stack(aff), T(g["aff_maps"]), rel=1e-5)
    single = ConstrainToBoundaryLossSingle.compute_affinities_single(T(g["aff_image"])[1].to(dev), 0.1, 5, 5)
    assert len(single) == 24 and tuple(single[0].shape) == (1, 16, 16)
    assert_close(torch.stack(single)[:, 0], T(g["aff_maps"])[:, 1, 0], rel=1e-5)


@pytest.mark.parametrize("shape,window,space,softmax,norm", [
    ((2, 2, 64, 96), 5, 0.0, True, 0), ((1, 3, 33, 70), 7, 3.0, True, 0), ((3, 2, 40, 40), 5, 5.0, False, 1),
    ((2, 4, 3, 50), 5, 5.0, True, 1), ((1, 2, 5, 4), 3, 0.0, False, 0)])
def test_pairwise_loss_vs_oracle(dev, shape, window, space, softmax, norm):
    import oracle
    from conftest import smooth_image
    from weaklysuperviseddl_amd import ops
    B, C, H, W = shape
    g = torch.Generator().manual_seed(H * W)
    img = smooth_image(B, H, W, 77)
    preds = torch.randn(shape, generator=g) * 2
    if not softmax:
        preds = F.softmax(preds, 1)
    pr = preds.clone().requires_grad_()
    ref = oracle.pairwise_affinity_loss(pr, img, window, 0.1, space, softmax, norm)
    wts = torch.linspace(0.5, 1.5, ref.numel()).view(ref.shape)
    (ref * wts).sum().backward()
    pd = preds.to(dev).requires_grad_()
    out = ops.pairwise_affinity_loss(pd, img.to(dev), window, 0.1, space, softmax, norm)
    assert_close(out, ref.detach(), what="loss")
    (out * wts.to(dev)).sum().backward()
    assert_close(pd.grad, pr.grad, what="grad")


def _mask_of(cam, thresh):
    return ((cam >= thresh) & (cam > 0)).to(torch.uint8)


def test_layercam_epilogue_vs_golden(dev, golden):
    """Identical activations / gradients in -> the CAM of the reference's own bodies out, BIT FOR BIT (the epilogue follows
    torch-CPU's operations and its order of summation, csrc/layercam_optim.hip), hence every mask index too: alpha = 1 (the
    default and every call site of the reference), 2 and 3.  alpha = 0.5 is torch.sqrt = MKL's vsSqrt, which is not
    correctly rounded: there the CAM is within one ulp and a mask pixel may differ only where the reference's value is
    within an ulp of the threshold.  The numbers of differing CAM values / mask pixels are reported (exact cases: 0 / 0)."""
    from conftest import report_line
    from weaklysuperviseddl_amd import ops
    g = golden("layercam")
    n_cam = n_mask = n_cam_sqrt = n_mask_sqrt = n_sqrt_vals = 0

    def compare(acts, grads, variant, a, ref, thr):
        nonlocal n_cam, n_mask, n_cam_sqrt, n_mask_sqrt, n_sqrt_vals
        cam, mask = ops.layercam_epilogue(acts, grads, (224, 224), a, variant, thresh=thr)
        cam, mask = cam.cpu(), mask.cpu()
        d_cam, d_mask = (cam != ref), (mask != _mask_of(ref, thr))
        if a == 0.5:
            ulp = ref.abs().clamp(min=1e-30) * 2.0 ** -22
            assert ((cam - ref).abs() <= ulp).all(), (variant, a, (cam - ref).abs().max().item())
            assert ((ref - thr).abs()[d_mask] <= 2.0 ** -22).all()
            n_cam_sqrt += d_cam.sum().item()
            n_mask_sqrt += d_mask.sum().item()
            n_sqrt_vals += ref.numel()
        else:
            n_cam += d_cam.sum().item()
            n_mask += d_mask.sum().item()

    for i in range(2):
        acts = [T(g[f"act_layer3_{i}"]).to(dev), T(g[f"act_layer4_{i}"]).to(dev)]
        grads = [T(g[f"grad_layer3_{i}"]).to(dev), T(g[f"grad_layer4_{i}"]).to(dev)]
        for variant, alphas in (("modular", (0.5, 1.0, 2.0)), ("notebook", (0.5, 2.0))):
            for a in alphas:
                for thr in (0.3, 0.5):
                    compare(acts, grads, variant, a, T(g[f"{variant}_cam_{i}_a{a}"]), thr)
    g = golden("layercam_wide")    # 300 / 600 channels: all cascade levels + left-over channels + the 4 scalar-column pixels
    acts = [T(g["act_layer3"]).to(dev), T(g["act_layer4"]).to(dev)]
    grads = [T(g["grad_layer3"]).to(dev), T(g["grad_layer4"]).to(dev)]
    for variant, a in (("modular", 1.0), ("modular", 3.0), ("notebook", 0.5)):
        compare(acts, grads, variant, a, T(g[f"{variant}_cam_a{a}"]), 0.3)
    report_line(f"layercam goldens (reference bodies, identical act/grad), alpha 1/2/3: CAM values differing {n_cam}, mask pixels "
                f"differing {n_mask}; alpha 0.5 (torch.sqrt = MKL vsSqrt, not correctly rounded): {n_cam_sqrt} of {n_sqrt_vals} CAM "
                f"values one ulp apart, mask pixels differing {n_mask_sqrt}")
    assert n_cam == 0 and n_mask == 0, (n_cam, n_mask)


def test_layercam_epilogue_full_size_vs_oracle(dev):
    """cfg1 shapes: layer3 (B,1024,14,14) + layer4 (B,2048,14,14), B=8 - bit-identical to the oracle (which equals the
    reference's bodies bit for bit on the fixtures, tests/test_oracle_golden.py), CAM and mask."""
    import oracle
    from weaklysuperviseddl_amd import ops
    g = torch.Generator().manual_seed(21)
    acts = [F.relu(torch.randn(8, c, 14, 14, generator=g)) for c in (1024, 2048)]
    grads = [torch.randn(8, c, 14, 14, generator=g) * 1e-3 for c in (1024, 2048)]
    from conftest import cpu_threads
    with cpu_threads(4):       # the thread count of the fixtures: torch-CPU's own order of summation depends on it
        ref = oracle.layercam_epilogue(acts, grads, (224, 224), 1.0, "modular")
    cam, mask = ops.layercam_epilogue([a.to(dev) for a in acts], [x.to(dev) for x in grads], (224, 224), 1.0,
                                      "modular", thresh=0.3)
    n_cam, n_mask = (cam.cpu() != ref).sum().item(), (mask.cpu() != _mask_of(ref, 0.3)).sum().item()
    from conftest import report_line
    report_line(f"layercam cfg1 size (8 x 1024/2048 x 14 x 14 vs oracle): CAM values differing {n_cam} of {ref.numel()}, mask pixels differing {n_mask}")
    assert n_cam == 0 and n_mask == 0
    # the reference's per-image loop (B = 1) on a 16-thread host sums the last four pixels by the cascade as well: option 0
    with cpu_threads(16):
        ref16 = torch.cat([oracle.layercam_epilogue([a[i:i + 1] for a in acts], [x[i:i + 1] for x in grads], (224, 224), 1.0, "modular")
                           for i in range(2)])
    ops.set_option("layercam_tail_mod", 0)
    try:
        cam16 = ops.layercam_epilogue([a[:2].to(dev) for a in acts], [x[:2].to(dev) for x in grads], (224, 224), 1.0, "modular")
    finally:
        ops.set_option("layercam_tail_mod", 32)
    assert torch.equal(cam16.cpu(), ref16)
    assert not torch.equal(ref16, ref[:2])      # the two orders do differ on this input


@pytest.mark.parametrize("shapes,out_hw,variant,alpha", [
    ([(37, 13, 13)], (224, 224), "modular", 1.0),                 # one layer, C % 4 != 0, 9 scalar-column pixels, scale 13/224
    ([(530, 11, 11), (64, 28, 28), (18, 7, 7)], (224, 224), "notebook", 2.0),   # three layers: the mean is a true division
    ([(600, 9, 9), (4097, 5, 5)], (100, 70), "modular", 2.0),     # every pixel of the 5x5 map in the scalar columns
    ([(96, 9, 9)], (50, 60), "modular", 1.0),                     # outH + outW <= 128: ATen's other bilinear kernel, ulp-close only
    ([(256, 32, 32), (16, 64, 64)], (256, 256), "modular", 3.0),  # no scalar columns
    ([(48, 24, 20)], (24, 20), "notebook", 1.0),                  # same-size interpolation (a copy)
    ([(300, 14, 14), (600, 14, 14)], (224, 224), "modular", 0.7), # general exponent: powf, within a few ulp only
])
def test_layercam_epilogue_odd_shapes_bit_identical_to_oracle(dev, shapes, out_hw, variant, alpha):
    import oracle
    from weaklysuperviseddl_amd import ops
    g = torch.Generator().manual_seed(sum(c for c, _, _ in shapes))
    B = 3
    acts = [F.relu(torch.randn(B, c, h, w, generator=g)) for c, h, w in shapes]
    grads = [torch.randn(B, c, h, w, generator=g) for c, h, w in shapes]
    from conftest import cpu_threads
    with cpu_threads(4):
        ref = oracle.layercam_epilogue(acts, grads, out_hw, alpha, variant)
    cam, mask = ops.layercam_epilogue([a.to(dev) for a in acts], [x.to(dev) for x in grads], out_hw, alpha, variant, thresh=0.4)
    if alpha == 0.7 or out_hw[0] + out_hw[1] <= 128:
        assert_close(cam, ref, rel=1e-6)
        return
    n = (cam.cpu() != ref).sum().item()
    assert n == 0, f"{n} of {ref.numel()} CAM values differ"
    assert torch.equal(mask.cpu(), _mask_of(ref, 0.4))


def test_adam_matches_torch(dev):
    from weaklysuperviseddl_amd.optim import FlatAdam
    g = torch.Generator().manual_seed(8)
    shapes = [(7, 3, 3, 3), (5,), (33, 17), (1,)]
    ps = [torch.randn(s, generator=g) for s in shapes]
    ref = [p.clone().requires_grad_() for p in ps]
    mine = [p.clone().to(dev).requires_grad_() for p in ps]
    opt_r = torch.optim.Adam(ref, lr=1e-2)
    opt_m = FlatAdam(mine, lr=1e-2)
    for step in range(5):
        grads = [torch.randn(s, generator=g) for s in shapes]
        opt_m.zero_grad()
        for p, gr in zip(ref, grads):
            p.grad = gr.clone()
        for p, gr in zip(mine, grads):
            p.grad.copy_(gr.to(dev))
        opt_r.step()
        opt_m.step()
    for a, b in zip(mine, ref):
        assert_close(a, b, rel=1e-5, what="adam params")


def test_softmax_kl(dev):
    from weaklysuperviseddl_amd import ops
    g = torch.Generator().manual_seed(6)
    x = torch.randn(1, 2, 12, 10, generator=g)
    s = F.softmax(torch.randn(1, 2, 12, 10, generator=g), 1)
    xr = x.clone().requires_grad_()
    xn = F.softmax(xr, 1)
    ref = F.kl_div((xn + 1e-8).log(), s, reduction="batchmean")
    ref.backward()
    xd = x.to(dev).requires_grad_()
    out = ops.kl_div_batchmean(ops.softmax_channels(xd), s.to(dev))
    assert_close(out, ref.detach(), rel=1e-4)
    out.backward()
    assert_close(xd.grad, xr.grad, rel=1e-4)


@pytest.mark.parametrize("shape,window", [((2, 2, 40, 70), 5), ((1, 3, 9, 33), 3), ((3, 2, 17, 23), 7), ((2, 2, 64, 64), 5)])
def test_pairwise_loss_with_cached_affinities_is_bit_identical(dev, shape, window):
    """wsdl_pairwise_cache + the cached kernel variant (what refine_pseudo_mask uses: the image is fixed over its steps)
    against the kernel that evaluates the affinities itself: same loss, same gradient, bit for bit, for both epilogues."""
    from conftest import smooth_image
    from weaklysuperviseddl_amd import ops
    B, C, H, W = shape
    img = smooth_image(B, H, W, 7).to(dev)
    preds = torch.randn(B, C, H, W, generator=torch.Generator().manual_seed(8)).to(dev)
    cache = ops.pairwise_cache(img, window, 0.1)
    assert tuple(cache.shape) == ((window * window - 1) // 2, B, H, W)
    for softmax, norm, space in ((True, 0, 0.0), (False, 1, 5.0), (True, 1, 0.0)):
        p1, p2 = preds.clone().requires_grad_(), preds.clone().requires_grad_()
        l1 = ops.pairwise_affinity_loss(p1, img, window, 0.1, space, softmax, norm)
        l2 = ops.pairwise_affinity_loss(p2, img, window, 0.1, space, softmax, norm, cache=cache)
        l1.sum().backward(), l2.sum().backward()
        assert torch.equal(l1, l2) and torch.equal(p1.grad, p2.grad), (softmax, norm, space)


def test_lovasz_softmax_vs_golden_and_oracle(dev, golden):
    """The device Lovasz-softmax against the vectors produced by the reference's own function bodies (loss and gradient;
    'present' with an absent class, 'all', per_image, an ignored label) and, at a size with a million pixels, against the
    oracle: loss to 1e-5; gradient where the sorted errors are distinct (inside a tie the reference's order is whatever
    torch.sort gave, ours the pixel index - the loss does not depend on it)."""
    import oracle
    from weaklysuperviseddl_amd import ops
    g = golden("lovasz")
    meta = json.loads(str(g["meta"]))
    for i, m in enumerate(meta):
        p = torch.from_numpy(g[f"lov{i}_probas"]).to(dev).requires_grad_()
        lab = torch.from_numpy(g[f"lov{i}_labels"]).to(dev)
        loss = ops.lovasz_softmax(p, lab, classes=m["classes"], per_image=m["per_image"], ignore=m["ignore"])
        loss.backward()
        assert abs(loss.item() - float(g[f"lov{i}_loss"])) <= 1e-5 * max(1.0, abs(float(g[f"lov{i}_loss"]))), (i, loss.item())
        ref = torch.from_numpy(g[f"lov{i}_grad"])
        assert (p.grad.cpu() - ref).abs().max().item() <= 1e-5 * ref.abs().max().item() + 1e-9, i
    gen = torch.Generator().manual_seed(5)
    logits = torch.randn(4, 2, 512, 512, generator=gen)
    labels = (torch.rand(4, 512, 512, generator=gen) > 0.6).long()
    pc = torch.softmax(logits, 1).requires_grad_()
    lo = oracle.lovasz_softmax(pc, labels)
    lo.backward()
    pd = torch.softmax(logits, 1).to(dev).requires_grad_()
    ld = ops.lovasz_softmax(pd, labels.to(dev))
    ld.backward()
    assert abs(ld.item() - lo.item()) <= 1e-5 * lo.item()
    # rank-dependent gradient: compare where no other pixel of the class has the same error
    err = (torch.nn.functional.one_hot(labels, 2).permute(0, 3, 1, 2).float() - pc.detach()).abs()
    for c in range(2):
        e = err[:, c].reshape(-1)
        _u, inv, cnt = torch.unique(e, return_inverse=True, return_counts=True)
        distinct = (cnt[inv] == 1).reshape(err[:, c].shape)
        a, b = pd.grad[:, c].cpu()[distinct], pc.grad[:, c][distinct]
        assert distinct.float().mean() > 0.5
        assert (a - b).abs().max().item() <= 2e-4 * b.abs().max().item()
    # two runs: bit-identical
    pd2 = torch.softmax(logits, 1).to(dev).requires_grad_()
    ld2 = ops.lovasz_softmax(pd2, labels.to(dev))
    ld2.backward()
    assert ld2.item() == ld.item() and torch.equal(pd2.grad, pd.grad)


def test_train_step_with_lovasz_softmax(dev):
    """loss_fn='lovasz_softmax' of the reference's train_segmentation_model: the step runs, the loss is finite and falls."""
    from weaklysuperviseddl_amd.TraditionalModel import build_segmentation_model, train_step
    from weaklysuperviseddl_amd.TraditionalModel.SegmentationModel import make_optimizer
    import bench
    torch.manual_seed(0)
    model = build_segmentation_model().to(dev).train()
    opt = make_optimizer(model, lr=1e-4)
    img, masks = bench.synthetic_batch(4, 64, 64, dev, 3)
    losses = [float(train_step(model, opt, img, masks, loss_fn="lovasz_softmax")) for _ in range(6)]
    assert all(np.isfinite(losses)) and min(losses[2:]) < losses[0], losses


# ------------------------------------------------------------------ keep_largest on the device (PsuedoMasks.py:15-21)
def _spiral(n):
    """ONE 1-pixel-wide arm winding inwards with a 1-pixel gap: the component with the longest label-propagation path."""
    m = np.zeros((n, n), np.uint8)
    r, c, dr, dc = 0, 0, 0, 1
    m[0, 0] = 1
    turns = 0
    while turns < 2:
        nr, nc, ar, ac = r + dr, c + dc, r + 2 * dr, c + 2 * dc
        ahead_taken = 0 <= ar < n and 0 <= ac < n and m[ar, ac]
        if 0 <= nr < n and 0 <= nc < n and not m[nr, nc] and not ahead_taken:
            r, c, turns = nr, nc, 0
            m[r, c] = 1
        else:
            dr, dc, turns = dc, -dr, turns + 1
    return m


def _keep_largest_cases():
    rng = np.random.RandomState(7)
    cases = {}
    for p in (0.1, 0.3, 0.45, 0.5, 0.6, 0.9):
        cases[f"noise{p}"] = (rng.rand(224, 224) < p).astype(np.uint8)
    yy, xx = np.mgrid[0:224, 0:224]
    blobs = np.zeros((224, 224), np.uint8)
    for cy, cx, r in ((60, 70, 40), (150, 160, 55), (30, 190, 20), (200, 30, 18)):
        blobs |= ((yy - cy) ** 2 + (xx - cx) ** 2 < r * r).astype(np.uint8)
    cases["blobs"] = blobs
    cases["blobs_noisy"] = blobs | (rng.rand(224, 224) < 0.05).astype(np.uint8)
    tie = np.zeros((224, 224), np.uint8)
    tie[100:110, 150:160] = 1           # three components of 100 pixels: the first in raster order must win ...
    tie[20:30, 200:210] = 1
    tie[20:30, 10:20] = 1               # ... this one (same first row as the one before it, further left)
    cases["tie"] = tie
    cases["empty"] = np.zeros((224, 224), np.uint8)
    cases["full"] = np.ones((224, 224), np.uint8)
    cases["values"] = (blobs * 255).astype(np.uint8)                      # any non-zero value is foreground
    cases["checker"] = ((yy + xx) % 2).astype(np.uint8)                   # diagonal neighbours connect (8-connectivity)
    cases["spiral"] = _spiral(224)
    comb = np.zeros((224, 224), np.uint8)                                 # teeth joined only by the LAST row
    comb[:, ::2] = 1
    comb[-1, :] = 1
    cases["comb"] = comb
    cases["ragged"] = (rng.rand(37, 53) < 0.55).astype(np.uint8)
    cases["one_pixel"] = np.ones((1, 1), np.uint8)
    cases["one_row"] = (rng.rand(1, 300) < 0.7).astype(np.uint8)
    cases["one_col"] = (rng.rand(300, 1) < 0.7).astype(np.uint8)
    cases["max_lds"] = (rng.rand(255, 257) < 0.52).astype(np.uint8)       # 65535 pixels: the last size with LDS labels
    cases["big"] = (rng.rand(300, 310) < 0.55).astype(np.uint8)           # 32-bit labels in the workspace
    cases["big_spiral"] = _spiral(320)
    return cases


def test_keep_largest_device_equals_the_reference_function(dev, golden):
    """Bit-exact: against the fixture made by the reference's own keep_largest (skimage, tests/golden/make_golden.py), the
    oracle on shapes the fixture does not hold (noise at the percolation threshold, ties, spirals, 8-connectivity, sizes
    either side of the LDS limit), batched (different masks in one launch) and in place."""
    import oracle
    from weaklysuperviseddl_amd import ops
    g = golden("keep_largest")
    names = sorted(k[3:] for k in g.files if k.startswith("in_"))
    assert names
    for n in names:
        m = np.ascontiguousarray(g["in_" + n]).astype(np.uint8)
        out = ops.keep_largest_batched(T(m).to(dev)).cpu().numpy()
        assert np.array_equal(out, g["out_" + n]), n
    cases = _keep_largest_cases()
    for n, m in cases.items():
        want = np.asarray(oracle.keep_largest(m)).astype(np.uint8)
        out = ops.keep_largest_batched(T(m).to(dev)).cpu().numpy()
        assert out.dtype == np.uint8 and np.array_equal(out, want), (n, int(out.sum()), int(want.sum()))
    # one launch over a batch of different 224 x 224 masks; bool input; in place through the C ABI
    batch = [k for k, v in cases.items() if v.shape == (224, 224)]
    stack = np.stack([cases[k] for k in batch])
    want = np.stack([np.asarray(oracle.keep_largest(cases[k])).astype(np.uint8) for k in batch])
    assert np.array_equal(ops.keep_largest_batched(T(stack).to(dev)).cpu().numpy(), want)
    assert np.array_equal(ops.keep_largest_batched(T(stack != 0).to(dev)).cpu().numpy(), want)
    from weaklysuperviseddl_amd._lib import lib
    from weaklysuperviseddl_amd.ops import _p, _stream, workspace, check
    buf = T(stack).to(dev)
    nb = lib().wsdl_keep_largest_workspace(*stack.shape)
    ws = workspace(nb, dev)
    check(lib().wsdl_keep_largest(_p(buf), _p(buf), *stack.shape, _p(ws), ws.numel(), _stream()))
    assert np.array_equal(buf.cpu().numpy(), want)
    # idempotent
    assert np.array_equal(ops.keep_largest_batched(T(want).to(dev)).cpu().numpy(), want)
    with pytest.raises(ops.WsdlError):
        ops.keep_largest_batched(torch.zeros(2, 8, 8, device=dev))         # float masks are refused, not cast
    with pytest.raises(ops.WsdlError):
        ops.keep_largest_batched(torch.zeros(2, 8, 8, dtype=torch.uint8))   # no CPU fallback


def test_stem_kernel_equals_the_generic_fp32_kernel(dev):
    """The 7x7 stride-2 stem on its own kernel (patch + weights in LDS) against the generic fp32 implicit-GEMM kernel: both exact
    fp32 products, different summation order - 1e-5 - with the eval-mode epilogue (scale, shift, ReLU, published amax)."""
    from weaklysuperviseddl_amd import ops
    g = torch.Generator().manual_seed(21)
    x = torch.randn(4, 3, 224, 224, generator=g).to(dev)
    w = (torch.randn(64, 3, 7, 7, generator=g) / 147 ** 0.5).to(dev)
    sc, sh = (torch.rand(64, generator=g) + 0.5).to(dev), torch.randn(64, generator=g).to(dev)
    wf, _ = ops.prep_weights(w)
    outs = []
    try:
        for on in (1, 0):
            ops.set_option("stem_kernel", on)
            y = ops.conv2d_fwd(x, wf, w.shape, 2, 3, 1)
            ye = ops.conv2d_fwd(x, wf, w.shape, 2, 3, 1, sc, sh, None, True)
            outs.append((y, ye, ops.amax_of(ye, True) if hasattr(ye, "_wsdl_amax") else None))
    finally:
        ops.set_option("stem_kernel", 1)
    (y1, ye1, a1), (y0, ye0, a0) = outs
    assert tuple(y1.shape) == (4, 64, 112, 112)
    assert rel_err(y1, y0) < 1e-5 and rel_err(ye1, ye0) < 1e-5
    ref = F.conv2d(x.cpu(), w.cpu(), None, 2, 3)
    assert rel_err(y1, ref) < 1e-4
    if a1 is not None:
        assert abs(a1.item() - ye1.abs().max().item()) <= 1e-6 * ye1.abs().max().item()


def test_prep_weights_multi_equals_the_single_launches(dev):
    """All split layouts in one launch (wsdl_conv2d_prep_weights_multi, a device table of descriptors) == one launch per
    convolution, byte for byte: 1x1 and 3x3, channel counts that are not multiples of the 32 x 32 staging tile."""
    from weaklysuperviseddl_amd import ops
    g = torch.Generator().manual_seed(31)
    shapes = [(64, 64, 3, 3), (256, 64, 1, 1), (128, 256, 1, 1), (48, 80, 3, 3), (512, 512, 3, 3), (2048, 512, 1, 1), (16, 16, 1, 1)]
    ws = [(torch.randn(s, generator=g) * (0.5 + i)).to(dev) for i, s in enumerate(shapes)]
    assert all(ops._both_split(w) for w in ws) and not ops._both_split(torch.empty(64, 3, 7, 7)) and not ops._both_split(torch.empty(2, 256, 1, 1))
    amax = ops.multi_amax(ws)
    sized = [ops.prep_weights(w, True, True, amax[i:i + 1]) for i, w in enumerate(ws)]      # (the buffers have unwritten padding:
    single = [ops.prep_weights(w, True, True, amax[i:i + 1], reuse=(torch.zeros_like(f), torch.zeros_like(d)))   # start both from zeros)
              for i, (w, (f, d)) in enumerate(zip(ws, sized))]
    multi = [(torch.zeros_like(f), torch.zeros_like(d)) for f, d in single]
    ops.prep_weights_multi([(w, f, d, amax[i:i + 1]) for i, (w, (f, d)) in enumerate(zip(ws, multi))])
    for (f1, d1), (f2, d2), s in zip(single, multi, shapes):
        assert torch.equal(f1.view(torch.int32), f2.view(torch.int32)) and torch.equal(d1.view(torch.int32), d2.view(torch.int32)), s
    # the table is cached per set of addresses: a second call with new weight values re-lays them out
    for w in ws:
        w.mul_(1.5)
    amax2 = ops.multi_amax(ws)
    single2 = [ops.prep_weights(w, True, True, amax2[i:i + 1], reuse=(torch.zeros_like(f), torch.zeros_like(d)))
               for i, (w, (f, d)) in enumerate(zip(ws, sized))]
    amax.copy_(amax2)
    ops.prep_weights_multi([(w, f, d, amax[i:i + 1]) for i, (w, (f, d)) in enumerate(zip(ws, multi))])
    for (f1, d1), (f2, d2) in zip(single2, multi):
        assert torch.equal(f1.view(torch.int32), f2.view(torch.int32)) and torch.equal(d1.view(torch.int32), d2.view(torch.int32))


def test_concat_into_with_producers_writing_in_place(dev):
    """ops.concat_into: BatchNorm kernels write two of three inputs straight into their channel slices of the buffer and publish
    into one shared amax slot; the third input is copied in and its maximum joins the slot.  == torch.cat, gradients are the
    slices, the slot holds the exact maximum."""
    from weaklysuperviseddl_amd import ops
    g = torch.Generator().manual_seed(33)
    B, H, W = 3, 8, 12
    xs = [torch.randn(B, c, H, W, generator=g).to(dev) for c in (16, 24)]
    third = (torch.randn(B, 8, H, W, generator=g) * 3).to(dev).requires_grad_()
    gam = [torch.rand(c, generator=g).to(dev) + 0.5 for c in (16, 24)]
    bet = [torch.randn(c, generator=g).to(dev) for c in (16, 24)]
    cat = torch.empty(B, 48, H, W, device=dev)
    slot = ops.amax_slot(dev)
    ys, ref, off = [], [], 0
    for x, ga, be, c in zip(xs, gam, bet, (16, 24)):
        rm, rv = torch.zeros(c, device=dev), torch.ones(c, device=dev)
        y, _, _ = ops.bn_train_fwd(x, ga, be, rm, rv, 0.1, 1e-5, None, True, out=ops._alias(cat[:, off:off + c]), amax_into=slot)
        yr, _, _ = ops.bn_train_fwd(x, ga, be, rm.clone(), rv.clone(), 0.1, 1e-5, None, True)
        ys.append(y)
        ref.append(yr)
        off += c
    out = ops.concat_into(cat, slot, ys + [third])
    want = torch.cat(ref + [third.detach()], dim=1)
    assert torch.equal(out, want) and out.data_ptr() == cat.data_ptr()
    assert abs(ops.amax_of(out, True).item() - want.abs().max().item()) <= 1e-6 * want.abs().max().item()
    dy = torch.randn(out.shape, generator=g).to(dev)
    out.backward(dy)
    assert torch.equal(third.grad, dy[:, 40:48])


@pytest.mark.parametrize("case", [(2, 128, 32, 32, 128, 3, 1, 1), (2, 128, 32, 32, 256, 3, 2, 2), (1, 256, 32, 32, 128, 3, 4, 4),
                                  (2, 128, 32, 32, 128, 3, 12, 12), (2, 256, 32, 32, 128, 1, 0, 1), (1, 128, 16, 64, 128, 3, 1, 1),
                                  (3, 128, 8, 96, 128, 3, 36, 36), (2, 1024, 32, 32, 256, 1, 0, 1)])
def test_wgrad_direct_fragments_equal_the_lds_staged_kernel(dev, case):
    """conv_wgrad_split16d_kernel (x fragments straight from global memory, masked at the image's left / right edge) against the
    LDS-staged kernel it replaces where OW % 32 == 0 and stride = 1: the same products in the same order - bit for bit - and
    against torch; dilations whose taps hang over the edge, rows of several chunks, taps that are dead altogether."""
    from weaklysuperviseddl_amd import ops
    B, Cin, H, W, Cout, k, pad, dil = case
    g = torch.Generator().manual_seed(41)
    x = torch.randn(B, Cin, H, W, generator=g)
    x[:, :, :, 0] *= 3.0          # edge columns carry weight: a wrong mask shows
    x[:, :, :, -1] *= 3.0
    dy = torch.randn(B, Cout, H, W, generator=g)
    wr = torch.zeros(Cout, Cin, k, k, requires_grad=True)
    F.conv2d(x, wr, None, 1, pad, dil).backward(dy)
    outs = []
    try:
        # 1 (default): the direct kernel where all taps are aligned (1x1, dilation 4 / 12 / 36), the LDS-staged kernel for dilation
        # 1 and 2; 0: LDS-staged everywhere
        for on in (1, 0):
            ops.set_option("wgrad_direct", on)
            outs.append(ops.conv2d_wgrad(x.to(dev), dy.to(dev), wr.shape, 1, pad, dil))
    finally:
        ops.set_option("wgrad_direct", 1)
    assert torch.equal(outs[0], outs[1]), (case, rel_err(outs[0], outs[1]))
    # dY read as fp32 and split while staged (default for 1x1 convolutions of few N tiles: the last case; 2 = every aligned launch)
    try:
        ops.set_option("wgrad_dyraw", 2)
        raw = ops.conv2d_wgrad(x.to(dev), dy.to(dev), wr.shape, 1, pad, dil)
        ops.set_option("wgrad_dyraw", 0)
        pre = ops.conv2d_wgrad(x.to(dev), dy.to(dev), wr.shape, 1, pad, dil)
    finally:
        ops.set_option("wgrad_dyraw", 1)
    assert torch.equal(raw, pre) and torch.equal(raw, outs[1]), case
    assert_close(outs[0], wr.grad, what=f"wgrad {case}")


@pytest.mark.parametrize("case", [(1024, 256, 1, 1, 1, 16, 4), (1024, 256, 1, 1, 1, 32, 2), (256, 256, 3, 1, 2, 32, 4),
                                  (256, 256, 3, 1, 12, 32, 2), (128, 256, 3, 2, 1, 32, 2)])
def test_wgrad_channel_scales_are_the_weight_gradients_range_guard(dev, case):
    """wsdl_set_option("wgrad_chan_scale", 1) - VERDICT r4 item 4c: the fp16x2 weight-gradient kernels (LDS-staged, direct-fragment,
    direct-fragment with dY split while staged) with one power-of-two scale per CHANNEL of x and of dY.  tools/wgrad_floor_probe.py's
    cases: one 2^20 outlier in x, in dY, in both.  With the per-tensor scale the both-outliers case leaves a row of dW - the row
    whose dY at the x outlier's pixel is 2^-29 of dY's maximum - at ~1e-3 of its own maximum; with per-channel scales every row
    and every column is <= 1e-4 (measured <= 2e-6).  On unit data the guard is as accurate as the default, and its
    results do not depend on which of the three kernels ran."""
    from weaklysuperviseddl_amd import ops
    Cin, Cout, k, s, d, H, B = case
    pad = (k // 2) * d if k > 1 else 0
    g = torch.Generator().manual_seed(78)
    worst = {}
    try:
        for what in ("none", "x", "dy", "both"):
            x = torch.randn(B, Cin, H, H, generator=g).to(dev)
            OH = ops.conv_out_hw(H, H, k, s, pad, d)[0]
            dy = torch.randn(B, Cout, OH, OH, generator=g).to(dev)
            if what in ("x", "both"):
                x[0, 3, H - 1, H - 1] = 2.0 ** 20
            if what in ("dy", "both"):
                dy[0, 5, 0, 0] = 2.0 ** 20
            ref = torch.nn.grad.conv2d_weight(x.double(), (Cout, Cin, k, k), dy.double(), s, pad, d)
            outs = []
            for opts in (dict(wgrad_chan_scale=1), dict(wgrad_chan_scale=1, wgrad_direct=0), dict(wgrad_chan_scale=1, wgrad_dyraw=0),
                         dict(wgrad_chan_scale=0)):
                for o, v in {**dict(wgrad_direct=1, wgrad_dyraw=1), **opts}.items():
                    ops.set_option(o, v)
                dw = ops.conv2d_wgrad(x, dy, (Cout, Cin, k, k), s, pad, d)
                e = (dw.double() - ref).abs()
                row = (e.amax(dim=(1, 2, 3)) / ref.abs().amax(dim=(1, 2, 3)).clamp_min(1e-30)).max().item()
                col = (e.amax(dim=(0, 2, 3)) / ref.abs().amax(dim=(0, 2, 3)).clamp_min(1e-30)).max().item()
                outs.append((dw, max(row, col)))
            guard, plain = max(o[1] for o in outs[:3]), outs[3][1]
            worst[what] = (guard, plain)
            assert guard <= 1e-4, (case, what, guard)
            assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][0], outs[2][0]), (case, what)
            if what == "none":
                assert guard <= 2.0 * plain + 1e-7, (case, guard, plain)
    finally:
        for o, v in dict(wgrad_chan_scale=0, wgrad_direct=1, wgrad_dyraw=1).items():
            ops.set_option(o, v)
    print(f"wgrad range guard {case}: worst row / column error, per-channel vs per-tensor scales: "
          + ", ".join(f"{k} {a:.1e} / {b:.1e}" for k, (a, b) in worst.items()))


@pytest.mark.parametrize("B,C,Co,H", [(16, 2048, 256, 32), (8, 2048, 256, 64), (16, 1024, 256, 32)])
def test_dgrad_multi_equals_the_chain_of_dgrads(dev, B, C, Co, H):
    """wsdl_conv2d_dgrad_multi (ASPP: the input gradient of the 1x1 and the three dilated 3x3 branches in ONE launch, the output
    tile accumulating over all sources' taps) against the chain of wsdl_conv2d_dgrad calls it replaces and against float64
    on a sample of output pixels; operands of very different magnitudes per source exercise the exact rescaling of the
    accumulators between sources."""
    from weaklysuperviseddl_amd import ops
    g = torch.Generator(device=dev).manual_seed(B + C)
    ks, dils = [1, 3, 3, 3], [1, 12, 24, 36]
    scales = [1.0, 3e-3, 40.0, 1e-5]
    ws = [torch.randn(Co, C, k, k, device=dev, generator=g) / (C * k * k) ** 0.5 for k in ks]
    dys = [torch.randn(B, Co, H, H, device=dev, generator=g) * sc for sc in scales]
    wds = [ops.prep_weights(w, False, True)[1] for w in ws]
    xshape = (B, C, H, H)
    if not ops.dgrad_multi_ok(4, xshape, Co):
        pytest.skip("geometry not served by the multi-source kernel")
    base = torch.randn(xshape, device=dev, generator=g)
    chain = base.clone()
    for dy, wd, w, d in zip(dys, wds, ws, dils):
        chain = ops.conv2d_dgrad(dy, wd, w.shape, xshape, 1, d * (w.shape[2] - 1) // 2, d, accumulate_into=chain)
    order = [1, 2, 3, 0]                                              # smallest dilation first (it decides the column bands)
    multi = ops.conv2d_dgrad_multi([dys[i] for i in order], [wds[i] for i in order], [tuple(ws[i].shape) for i in order],
                                   [dils[i] for i in order], xshape, accumulate_into=base.clone())
    plain = ops.conv2d_dgrad_multi([dys[i] for i in order], [wds[i] for i in order], [tuple(ws[i].shape) for i in order],
                                   [dils[i] for i in order], xshape)
    torch.cuda.synchronize()
    scale = chain.abs().max().item()
    assert ((multi - chain).abs().max().item()) <= 2e-5 * scale
    assert ((plain + base - chain).abs().max().item()) <= 2e-5 * scale
    # float64 on a few output positions (all channels): dx[b, :, y, x] = sum_i sum_taps W_i[:, :, ky, kx]^T dy_i[b, :, y + pad - ky d, ...]
    ref_pts = [(0, 0, 0), (B - 1, H - 1, H - 1), (1, 13, 7), (2, H // 2, H // 2), (3, 5, H - 2)]
    for b, y, x in ref_pts:
        acc = base[b, :, y, x].double().cpu()
        for dy, w, d in zip(dys, ws, dils):
            k = w.shape[2]
            pad = d * (k - 1) // 2
            for ky in range(k):
                for kx in range(k):
                    oy, ox = y + pad - ky * d, x + pad - kx * d
                    if 0 <= oy < H and 0 <= ox < H:
                        acc += w[:, :, ky, kx].double().cpu().t() @ dy[b, :, oy, ox].double().cpu()
        err = (multi[b, :, y, x].double().cpu() - acc).abs().max().item()
        assert err <= 1e-5 * max(acc.abs().max().item(), 1e-30), (b, y, x, err)


@pytest.mark.parametrize("B,C,Co,H", [(16, 2048, 256, 32), (8, 2048, 256, 64), (4, 512, 256, 32)])
def test_fwd_group_equals_the_single_launches(dev, B, C, Co, H):
    """wsdl_conv2d_fwd_group (ASPP's four branch convolutions as ONE grid, tiles of more than two taps cut into K slices)
    against one wsdl_conv2d_fwd launch per branch: the same products in another order of summation."""
    from weaklysuperviseddl_amd import ops
    g = torch.Generator(device=dev).manual_seed(B * 3 + C)
    ks, dils = [1, 3, 3, 3], [1, 12, 24, 36]
    x = torch.randn(B, C, H, H, device=dev, generator=g)
    ws = [torch.randn(Co, C, k, k, device=dev, generator=g) / (C * k * k) ** 0.5 for k in ks]
    wfs = [ops.prep_weights(w, True, False)[0] for w in ws]
    if not ops.fwd_group_ok(4, tuple(x.shape), Co):
        pytest.skip("geometry not served by the grouped kernel")
    single = [ops.conv2d_fwd(x, wf, w.shape, 1, d * (w.shape[2] - 1) // 2, d) for wf, w, d in zip(wfs, ws, dils)]
    grouped = ops.conv2d_fwd_group(x, wfs, [tuple(w.shape) for w in ws], dils)
    again = ops.conv2d_fwd_group(x, wfs, [tuple(w.shape) for w in ws], dils)
    torch.cuda.synchronize()
    for a, b, c in zip(single, grouped, again):
        assert ((a - b).abs().max() / a.abs().max()).item() < 1e-5       # K up to 18432 summed in another order
        assert torch.equal(b, c)                                       # fixed-order reduce: bitwise reproducible


def test_range_sentinel_flags_tensors_beyond_the_safe_range(dev):
    """The BatchNorm kernels publish, besides max|tensor|, the smallest non-zero channel maximum (range sentinel);
    wsdl_range_check turns a step's pairs into "some tensor spans more than 2^25".  Unit data: a few bits of spread, no flag.
    Channels graded by 2^30 (gamma falling by 2^30 across the channels: the forward output and the input gradient both carry
    it): flagged, worst spread ~2^30 - the signal that selects conv_arith = 2.  (Grading the IMAGES of a batch does not survive
    a BatchNorm: its backward mixes the images of a channel through the batch statistics - checked here too.)"""
    from weaklysuperviseddl_amd import ops
    g = torch.Generator(device=dev).manual_seed(9)
    B, C, H = 4, 256, 32
    x = torch.randn(B, C, H, H, device=dev, generator=g)
    gamma, beta = torch.rand(C, device=dev, generator=g) + 0.5, torch.randn(C, device=dev, generator=g) * 0.1
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)

    def run(dy, ga, be):
        ops.reset_amax_pool(dev)
        y, mean, invstd = ops.bn_train_fwd(x, ga, be, rm.clone(), rv.clone(), 0.1, 1e-5, relu=True)
        dx, _, _, _ = ops.bn_train_bwd(x, dy, None, ga, mean, invstd, True, False, beta=be)
        ops.range_check(dev)
        torch.cuda.synchronize()
        return ops.range_status(dev), y, dx

    dy = torch.randn(B, C, H, H, device=dev, generator=g)
    st, _, _ = run(dy, gamma, beta)
    assert st["pairs_seen"] >= 2 and not st["exceeded"] and st["worst_log2"] <= 14, st
    grade_b = torch.tensor([2.0 ** (-30.0 * b / (B - 1)) for b in range(B)], device=dev).view(B, 1, 1, 1)
    st, _, _ = run(dy * grade_b, gamma, beta)
    assert not st["exceeded"], st                                   # the backward's batch statistics mix the images
    grade_c = torch.tensor([2.0 ** (-30.0 * c / (C - 1)) for c in range(C)], device=dev)
    st, y, dx = run(dy, gamma * grade_c, beta * grade_c)
    assert st["exceeded"] and st["pairs_over_limit"] == 2 and 25 < st["worst_log2"] <= 36, st
    # the published maximum is still the tensor's (the slot's first float), whatever the second one holds
    for t in (y, dx):
        assert abs(float(t._wsdl_amax) - t.abs().max().item()) <= 1e-6 * t.abs().max().item()
    ops.reset_amax_pool(dev)


def test_batchnorm_with_several_workgroups_per_channel(dev):
    """"bn_coop": the 64-channel layers' BatchNorm runs four workgroups per channel that hand their partial sums over and add
    them in one order - the results equal the one-workgroup-per-channel kernels' (partial sums are doubles: to the last bit on
    these shapes, asserted to 1e-6), are bitwise reproducible, leave their counters zeroed (a second launch works), and odd
    geometries fall back."""
    from weaklysuperviseddl_amd import ops
    g = torch.Generator(device=dev).manual_seed(12)
    for B, C, H, res in ((16, 64, 64, False), (4, 64, 32, True), (3, 48, 20, False), (16, 32, 64, False)):
        x = torch.randn(B, C, H, H, device=dev, generator=g)
        dy = torch.randn(B, C, H, H, device=dev, generator=g)
        r = torch.randn(B, C, H, H, device=dev, generator=g) if res else None
        gamma, beta = torch.rand(C, device=dev, generator=g) + 0.5, torch.randn(C, device=dev, generator=g) * 0.1
        out = {}
        for coop in (0, 64):
            ops.set_option("bn_coop", coop)
            try:
                runs = []
                for _ in range(2):
                    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
                    y, mean, invstd, bits = ops.bn_train_fwd(x, gamma, beta, rm, rv, 0.1, 1e-5, r, True, want_mask=True, mask_if=res)
                    dx, dgam, dbet, dres = ops.bn_train_bwd(x, dy, y if (res and bits is None) else None, gamma, mean, invstd, True, res,
                                                            beta=None if res else beta, relu_mask=bits)
                    runs.append([t.clone() for t in (y, mean, invstd, rm, rv, dx, dgam, dbet)] + ([dres.clone()] if res else []))
                torch.cuda.synchronize()
                assert all(torch.equal(a, b) for a, b in zip(*runs))
                out[coop] = runs[0]
            finally:
                ops.set_option("bn_coop", 0)              # the default
        for a, b in zip(out[0], out[64]):
            assert torch.isfinite(b).all()
            assert ((a - b).abs().max() / (a.abs().max() + 1e-30)).item() < 1e-6
    cnt = ops.coop_counters(dev)
    assert cnt is not None and int(cnt.abs().sum()) == 0


@pytest.mark.parametrize("case", [(4, 256, 32, 256, 3, 2), (2, 256, 32, 128, 3, 4), (4, 128, 32, 128, 3, 1), (2, 1024, 32, 256, 1, 1)])
def test_bn_backward_writes_the_weight_gradients_dy_operand_and_channel_maxima(dev, case):
    """Round 6: the channel-resident BatchNorm kernels publish one maximum per CHANNEL (forward: of y, backward: of dx) and
    the backward writes dx a second time as the fp16 (high, low) rows the producing convolution's weight gradient reads, scaled
    per channel (wsdl_bn_train_bwd chan_amax / dy_presplit; wsdl_conv2d_wgrad_ex).  What that replaces: the per-tensor scale +
    dy_split16 pass of round 5, or - same arithmetic - the "wgrad_chan_scale" pre-pass.  So: (i) the channel maxima are exact;
    (ii) the weight gradient from (published maxima, pre-split rows) equals the one from the pre-pass BIT FOR BIT; (iii) it
    is as close to float64 as the per-tensor form or closer; (iv) a channel 2^-30 below the others keeps its accuracy."""
    from weaklysuperviseddl_amd import ops
    B, Cin, H, Cout, k, d = case
    pad = (k // 2) * d
    g = torch.Generator(device=dev).manual_seed(sum(case))
    x = torch.relu(torch.randn(B, Cin, H, H, device=dev, generator=g))
    conv_out = torch.randn(B, Cout, H, H, device=dev, generator=g)
    dy = torch.randn(B, Cout, H, H, device=dev, generator=g)
    gamma = torch.rand(Cout, device=dev, generator=g) + 0.5
    gamma[::7] *= 2.0 ** -30                                    # graded channels: the case per-tensor scales lose bits on
    beta = torch.randn(Cout, device=dev, generator=g) * 0.1
    rm, rv = torch.zeros(Cout, device=dev), torch.ones(Cout, device=dev)
    wshape = (Cout, Cin, k, k)
    # forward: the per-channel maxima of y
    xin = torch.randn(B, Cin, H, H, device=dev, generator=g)
    gx, bx = torch.rand(Cin, device=dev, generator=g) + 0.5, torch.randn(Cin, device=dev, generator=g) * 0.1
    y, _m, _i = ops.bn_train_fwd(xin, gx, bx, torch.zeros(Cin, device=dev), torch.ones(Cin, device=dev), 0.1, 1e-5, relu=True)
    assert getattr(y, "_wsdl_camax", None) is not None, "the forward kernel of this shape is channel-resident"
    assert torch.equal(y._wsdl_camax, y.abs().amax(dim=(0, 2, 3)))
    _y, mean, invstd = ops.bn_train_fwd(conv_out, gamma, beta, rm, rv, 0.1, 1e-5, relu=True)
    psb = ops.wgrad_presplit_bytes(tuple(x.shape), wshape, 1, pad, d)
    assert (psb > 0) == (k == 3), psb                           # the 1x1 case reads dY as fp32 (few N tiles): nothing to pre-split
    dx, _, _, _ = ops.bn_train_bwd(conv_out, dy, None, gamma, mean, invstd, True, False, beta=beta, presplit_bytes=psb)
    cam = dx._wsdl_camax
    assert torch.equal(cam, dx.abs().amax(dim=(0, 2, 3)))
    assert (getattr(dx, "_wsdl_presplit", None) is not None) == (psb > 0)
    ref = torch.nn.grad.conv2d_weight(x.double().cpu(), wshape, dx.double().cpu(), 1, pad, d)

    def err(dw):      # worst row of dW against float64, relative to the row's own maximum
        e = (dw.double().cpu() - ref).abs().amax(dim=(1, 2, 3))
        return (e / ref.abs().amax(dim=(1, 2, 3)).clamp_min(1e-300)).max().item()
    ops.launch_trace(True)
    try:
        ops.last_launches()
        dw_new = ops.conv2d_wgrad(x, dx, wshape, 1, pad, d, x_camax=None, dy_camax=cam, dy_presplit=getattr(dx, "_wsdl_presplit", None))
        trace = ops.last_launches()
    finally:
        ops.launch_trace(False)
    assert ("pre-split by its producer" in trace) == (psb > 0), trace
    assert "dy_split16" not in trace and "cs=1" in trace, trace
    plain = torch.empty_like(dx).copy_(dx)                      # the same values without the producer's attributes
    dw_tensor = ops.conv2d_wgrad(x, plain, wshape, 1, pad, d)
    ops.set_option("wgrad_chan_scale", 1)
    try:
        dw_prepass = ops.conv2d_wgrad(x, plain, wshape, 1, pad, d)
    finally:
        ops.set_option("wgrad_chan_scale", 0)
    # the pre-pass takes per-channel maxima of BOTH operands, the new path here only of dY: compare like with like
    dw_new_both = ops.conv2d_wgrad(x, dx, wshape, 1, pad, d, x_camax=x.abs().amax(dim=(0, 2, 3)).contiguous(), dy_camax=cam,
                                   dy_presplit=getattr(dx, "_wsdl_presplit", None))
    assert torch.equal(dw_new_both, dw_prepass)
    e_new, e_tensor = err(dw_new), err(dw_tensor)
    assert e_new <= 2e-6, e_new                                  # every row, the 2^-30 ones too
    assert e_new <= 1.5 * e_tensor + 1e-7, (e_new, e_tensor)
    from conftest import report_line
    report_line(f"weight gradient {case}: worst row vs float64 - per-channel scales from the BatchNorm backward {e_new:.1e}, per-tensor scale "
                f"{e_tensor:.1e} (every 7th channel of dY 2^-30 below the rest)")
