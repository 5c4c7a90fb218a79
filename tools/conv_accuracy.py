"""Accuracy of the two convolution arithmetic paths against an fp64 reference (GPU, through the C ABI).

For a handful of DeepLabV3-R50 shapes: forward, input-gradient and weight-gradient results of
  * the fp32-MFMA kernels            (wsdl_set_option("conv_split", 0)), and
  * the bf16x3-split kernels         (conv_split = 1: six bf16 MFMAs per product, conv_split.h)
are compared with torch's fp64 convolution on the same device.  Reported: max |err| / max |ref| and
rms err / rms ref.  The split path must not be worse than the fp32 path by more than a small factor.

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


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    print(f"{'shape':34s} {'pass':6s} {'fp32 max':>10s} {'fp32 rms':>10s} {'split max':>10s} {'split rms':>10s}")
    worst = 0.0
    for Cin, Cout, k, s, d, H, B in SHAPES:
        pad = (k // 2) * d if k > 1 else 0
        x = torch.randn(B, Cin, H, H, device=dev)
        w = torch.randn(Cout, Cin, k, k, device=dev) / (Cin * k * k) ** 0.5
        ref = F.conv2d(x.double(), w.double(), None, s, pad, d)
        dy = torch.randn_like(ref, dtype=torch.float32)
        ref_dx = torch.nn.grad.conv2d_input(x.shape, w.double(), dy.double(), s, pad, d)
        ref_dw = torch.nn.grad.conv2d_weight(x.double(), w.shape, dy.double(), s, pad, d)
        res = {}
        for mode in (0, 1):
            ops.set_option("conv_split", mode)
            ops.set_option("wgrad_split", mode)
            wf, wd = ops.prep_weights(w)
            y = ops.conv2d_fwd(x, wf, w.shape, s, pad, d)
            dx = ops.conv2d_dgrad(dy, wd, w.shape, x.shape, s, pad, d)
            dw = ops.conv2d_wgrad(x, dy, w.shape, s, pad, d)
            res[mode] = (errs(y, ref), errs(dx, ref_dx), errs(dw, ref_dw))
        name = f"{Cin}->{Cout} k{k} s{s} d{d} {H}x{H} B{B}"
        for i, ps in enumerate(("fwd", "dgrad", "wgrad")):
            f32, sp = res[0][i], res[1][i]
            print(f"{name:34s} {ps:6s} {f32[0]:10.2e} {f32[1]:10.2e} {sp[0]:10.2e} {sp[1]:10.2e}")
            worst = max(worst, sp[1] / f32[1])
    print(f"worst split/fp32 rms-error ratio: {worst:.2f}")


if __name__ == "__main__":
    main()
