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
    a = ap.parse_args()
    for kv in [x for x in a.opt.split(",") if x]:
        k, v = kv.split("=")
        ops.set_option(k, v)
    Cin, Cout, k, s, d, H, B = [int(v) for v in a.shape.split(",")]
    dev = torch.device("cuda:0")
    pad = (k // 2) * d if k > 1 else 0
    x = torch.randn(B, Cin, H, H, device=dev)
    w = torch.randn(Cout, Cin, k, k, device=dev) * 0.05
    wf, wd = ops.prep_weights(w)
    OH, OW = ops.conv_out_hw(H, H, k, s, pad, d)
    dy = torch.randn(B, Cout, OH, OW, device=dev)
    if a.zeros:
        x.zero_(); w.zero_(); dy.zero_()
        wf, wd = ops.prep_weights(w)
    for _ in range(a.reps):
        if a.which == "fwd":
            ops.conv2d_fwd(x, wf, w.shape, s, pad, d)
        elif a.which == "dgrad":
            ops.conv2d_dgrad(dy, wd, w.shape, x.shape, s, pad, d)
        else:
            ops.conv2d_wgrad(x, dy, w.shape, s, pad, d)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
