"""Accuracy of the convolution arithmetic paths against an fp64 reference (GPU, through the C ABI).

For a handful of DeepLabV3-R50 shapes: forward, input-gradient and weight-gradient results of
  * the fp32-MFMA kernels            (wsdl_set_option("conv_split", 0): exact fp32 fma chains),
  * the bf16x3-split kernels         (conv_arith = 0: six bf16 MFMAs per product), and
  * the fp16x2-split kernels         (conv_arith = 1, the default: three fp16 MFMAs per product, per-tensor
                                      power-of-two scales - conv_split.h)
are compared with torch's fp64 convolution on the same device.  Reported: rms err / rms ref (and max |err| / max |ref|
for the default path).  Inputs: unit-variance noise, and a "wide" variant whose channels / pixels span seven decades
and whose gradients are tiny (1e-6) - what the scales have to cope with.  A split path must not be worse than the
fp32 path by more than a small factor.

    python tools/conv_accuracy.py
"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from weaklysuperviseddl_amd import ops  # noqa: E402

SHAPES = [  # Cin, Cout, k, stride, dil, H, B
    (64, 64, 3, 1, 1, 64, 4),
    (256, 64, 1, 1, 1, 64, 4),
    (512, 512, 3, 1, 2, 32, 4),
    (2048, 256, 3, 1, 12, 32, 2),
    (1024, 2048, 1, 1, 1, 32, 2),
    (128, 128, 3, 2, 1, 64, 4),
    (256, 512, 1, 2, 1, 16, 4),     # small maps: split-K launches
    (128, 128, 3, 2, 1, 16, 4),
    (256, 128, 1, 1, 1, 16, 4),
    (128, 512, 1, 1, 1, 8, 4),
    (2048, 256, 3, 1, 4, 14, 2),
    (256, 1024, 1, 1, 1, 32, 2),    # role-swapped weight gradient
]


def errs(a, ref):
    d = (a.double() - ref)
    return (d.abs().max() / ref.abs().max()).item(), (d.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()


MODES = (("fp32", dict(conv_split=0, wgrad_split=0, conv_arith=1)),
         ("bf16x3", dict(conv_split=1, wgrad_split=1, conv_arith=0)),
         ("fp16x2", dict(conv_split=1, wgrad_split=1, conv_arith=1)),      # the default
         # the range guard: low piece at 2^11, second accumulator (forward / input gradient; the weight gradient is the default's)
         ("fp16x2-alt", dict(conv_split=1, wgrad_split=1, conv_arith=2)))


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    print(f"{'shape':34s} {'data':5s} {'pass':6s} {'fp32 rms':>10s} {'bf16x3 rms':>11s} {'fp16x2 rms':>11s} {'fp16x2 max':>11s} {'fp16x2s rms':>11s}")
    worst = {"bf16x3": 0.0, "fp16x2": 0.0, "fp16x2-alt": 0.0}
    for Cin, Cout, k, s, d, H, B in SHAPES:
        pad = (k // 2) * d if k > 1 else 0
        for data in ("unit", "wide"):
            x = torch.randn(B, Cin, H, H, device=dev)
            w = torch.randn(Cout, Cin, k, k, device=dev) / (Cin * k * k) ** 0.5
            if data == "wide":
                x = torch.relu(x) * torch.logspace(-4, 3, Cin, device=dev).view(1, Cin, 1, 1)
                w = w * torch.logspace(-2, 2, Cout, device=dev).view(Cout, 1, 1, 1)
            ref = F.conv2d(x.double(), w.double(), None, s, pad, d)
            dy = torch.randn_like(ref, dtype=torch.float32)
            if data == "wide":
                dy = dy * 1e-6 * torch.logspace(-3, 3, dy.shape[-1], device=dev)
            ref_dx = torch.nn.grad.conv2d_input(x.shape, w.double(), dy.double(), s, pad, d)
            ref_dw = torch.nn.grad.conv2d_weight(x.double(), w.shape, dy.double(), s, pad, d)
            res = {}
            for mode, opts in MODES:
                for o, v in opts.items():
                    ops.set_option(o, v)
                wf, wd = ops.prep_weights(w)
                y = ops.conv2d_fwd(x, wf, w.shape, s, pad, d)
                dx = ops.conv2d_dgrad(dy, wd, w.shape, x.shape, s, pad, d)
                dw = ops.conv2d_wgrad(x, dy, w.shape, s, pad, d)
                res[mode] = (errs(y, ref), errs(dx, ref_dx), errs(dw, ref_dw))
            name = f"{Cin}->{Cout} k{k} s{s} d{d} {H}x{H} B{B}"
            for i, ps in enumerate(("fwd", "dgrad", "wgrad")):
                f32, b3, h2, ha = res["fp32"][i], res["bf16x3"][i], res["fp16x2"][i], res["fp16x2-alt"][i]
                print(f"{name:34s} {data:5s} {ps:6s} {f32[1]:10.2e} {b3[1]:11.2e} {h2[1]:11.2e} {h2[0]:11.2e} {ha[1]:11.2e}")
                worst["bf16x3"] = max(worst["bf16x3"], b3[1] / f32[1])
                worst["fp16x2"] = max(worst["fp16x2"], h2[1] / f32[1])
                worst["fp16x2-alt"] = max(worst["fp16x2-alt"], ha[1] / f32[1])
    for o, v in MODES[2][1].items():
        ops.set_option(o, v)
    print("worst split/fp32 rms-error ratio: bf16x3 %.2f, fp16x2 %.2f (default), fp16x2s (conv_arith = 2) %.2f"
          % (worst["bf16x3"], worst["fp16x2"], worst["fp16x2-alt"]))


if __name__ == "__main__":
    main()
