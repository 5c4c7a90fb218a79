"""Run ONE convolution pass a few times (for rocprofv3 --pmc runs on a single kernel).

    python tools/one_conv.py --shape 512,512,3,1,4,32,16 --pass wgrad [--reps 3] [--opt name=value,...]
shape = Cin,Cout,k,stride,dil,H,B
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from weaklysuperviseddl_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="512,512,3,1,4,32,16")
    ap.add_argument("--pass", dest="which", default="fwd")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--opt", default="")
    ap.add_argument("--zeros", action="store_true", help="all-zero operands (power / clock experiment)")
    ap.add_argument("--plain", action="store_true", help="weight gradient as a stand-alone call (per-tensor scales, dy_split16, its own reduce launch)")
    a = ap.parse_args()
    for kv in [x for x in a.opt.split(",") if x]:
        k, v = kv.split("=")
        ops.set_option(k, v)
    dev = torch.device("cuda:0")
    if a.shape.startswith("aspp"):
        # ASPP's four branch convolutions as the step runs them (round 5): one grouped forward launch / one multi-source
        # input-gradient launch.  --shape aspp[,B[,H]]
        parts = a.shape.split(",")
        B, H = (int(parts[1]) if len(parts) > 1 else 16), (int(parts[2]) if len(parts) > 2 else 32)
        ks, dils = [1, 3, 3, 3], [1, 12, 24, 36]
        x = torch.randn(B, 2048, H, H, device=dev)
        ws = [torch.randn(256, 2048, k, k, device=dev) * 0.02 for k in ks]
        preps = [ops.prep_weights(w) for w in ws]
        dys = [torch.randn(B, 256, H, H, device=dev) for _ in ks]
        if a.zeros:
            x.zero_()
            ws = [w.zero_() for w in ws]
            dys = [d.zero_() for d in dys]
            preps = [ops.prep_weights(w) for w in ws]
        shapes = [tuple(w.shape) for w in ws]
        order = [1, 2, 3, 0]
        for _ in range(a.reps):
            if a.which == "fwd":
                ops.conv2d_fwd_group(x, [p[0] for p in preps], shapes, dils)
            else:
                ops.conv2d_dgrad_multi([dys[i] for i in order], [preps[i][1] for i in order], [shapes[i] for i in order],
                                       [dils[i] for i in order], tuple(x.shape))
        torch.cuda.synchronize()
        return
    Cin, Cout, k, s, d, H, B = [int(v) for v in a.shape.split(",")]
    pad = (k // 2) * d if k > 1 else 0
    x = torch.randn(B, Cin, H, H, device=dev)
    w = torch.randn(Cout, Cin, k, k, device=dev) * 0.05
    wf, wd = ops.prep_weights(w)
    OH, OW = ops.conv_out_hw(H, H, k, s, pad, d)
    dy = torch.randn(B, Cout, OH, OW, device=dev)
    if a.zeros:
        x.zero_(); w.zero_(); dy.zero_()
        wf, wd = ops.prep_weights(w)
    # the weight gradient AS THE TRAINING STEP RUNS IT (round 6): its dY operand comes from a BatchNorm backward, which publishes the
    # per-channel maxima and - where the launch reads them - the pre-split rows (no dy_split16 pass); the slab reduction is deferred
    # (in the step: one launch for all layers; here one launch for this layer - an upper bound of its share)
    cam = pre = None
    if a.which == "wgrad" and not a.plain:
        gamma, beta = torch.rand(Cout, device=dev) + 0.5, torch.zeros(Cout, device=dev)
        conv_out = torch.randn(B, Cout, OH, OW, device=dev)
        _y, mean, invstd = ops.bn_train_fwd(conv_out, gamma, beta, torch.zeros(Cout, device=dev), torch.ones(Cout, device=dev), 0.1, 1e-5, relu=True)
        psb = ops.wgrad_presplit_bytes(tuple(x.shape), tuple(w.shape), s, pad, d)
        if a.zeros:
            dy.zero_()
        dyb, _, _, _ = ops.bn_train_bwd(conv_out, dy, None, gamma, mean, invstd, True, False, beta=beta, presplit_bytes=psb)
        dy = dyb
        cam, pre = getattr(dy, "_wsdl_camax", None), getattr(dy, "_wsdl_presplit", None)
    dw = torch.zeros_like(w)
    for _ in range(a.reps):
        if a.which == "fwd":
            ops.conv2d_fwd(x, wf, w.shape, s, pad, d)
        elif a.which == "dgrad":
            ops.conv2d_dgrad(dy, wd, w.shape, x.shape, s, pad, d)
        elif a.plain:
            ops.conv2d_wgrad(x, dy, w.shape, s, pad, d)
        else:
            ops.conv2d_wgrad(x, dy, w.shape, s, pad, d, out=dw, defer=True, dy_camax=cam, dy_presplit=pre)
            ops.flush_wgrad_reduces(dev)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
