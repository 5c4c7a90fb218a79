#!/usr/bin/env python3
"""Per-shape timing of the conv kernels on the unique DeepLabV3-R50 shapes at 256x256.

    python tools/conv_shapes_bench.py [--batch 16] [--reps 10] [--only fwd|dgrad|wgrad]
Prints one line per (shape, pass): microseconds; nominal TFLOP/s (dense FLOPs, padding taps counted - an fp32-equivalent
rate, NOT a utilisation); executed TFLOP/s (padding-only K chunks the kernel skips left out); and `frac` = the MFMA work
really issued against the peak of the MFMA type that ran it: executed x (16-bit products per fp32 product: 3 for the
fp16x2 split kernels, 6 for bf16x3) / 2516.6 for the split kernels, executed / 157.3 for the fp32-MFMA kernels.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from weaklysuperviseddl_amd import ops  # noqa: E402

PEAK32, PEAK16 = 157.3, 2516.6
# (count, Cin, Cout, k, stride, dil, Hin, name)
SHAPES = [
    (1, 3, 64, 7, 2, 1, 256, "conv1"),
    (1, 64, 64, 1, 1, 1, 64, "l1.0.conv1"),
    (3, 64, 64, 3, 1, 1, 64, "l1.conv2"),
    (4, 64, 256, 1, 1, 1, 64, "l1.conv3/ds"),
    (2, 256, 64, 1, 1, 1, 64, "l1.conv1"),
    (1, 256, 128, 1, 1, 1, 64, "l2.0.conv1"),
    (1, 128, 128, 3, 2, 1, 64, "l2.0.conv2 s2"),
    (4, 128, 512, 1, 1, 1, 32, "l2.conv3"),
    (1, 256, 512, 1, 2, 1, 64, "l2.0.ds s2"),
    (3, 512, 128, 1, 1, 1, 32, "l2.conv1"),
    (3, 128, 128, 3, 1, 1, 32, "l2.conv2"),
    (1, 512, 256, 1, 1, 1, 32, "l3.0.conv1"),
    (2, 256, 256, 3, 1, 1, 32, "l3.0.conv2/head3x3"),
    (6, 256, 1024, 1, 1, 1, 32, "l3.conv3"),
    (1, 512, 1024, 1, 1, 1, 32, "l3.0.ds"),
    (5, 1024, 256, 1, 1, 1, 32, "l3.conv1"),
    (5, 256, 256, 3, 1, 2, 32, "l3.conv2 d2"),
    (1, 1024, 512, 1, 1, 1, 32, "l4.0.conv1"),
    (1, 512, 512, 3, 1, 2, 32, "l4.0.conv2 d2"),
    (3, 512, 2048, 1, 1, 1, 32, "l4.conv3"),
    (1, 1024, 2048, 1, 1, 1, 32, "l4.0.ds"),
    (2, 2048, 512, 1, 1, 1, 32, "l4.conv1"),
    (2, 512, 512, 3, 1, 4, 32, "l4.conv2 d4"),
    (1, 2048, 256, 1, 1, 1, 32, "aspp 1x1"),
    (1, 2048, 256, 3, 1, 12, 32, "aspp d12"),
    (1, 2048, 256, 3, 1, 24, 32, "aspp d24"),
    (1, 2048, 256, 3, 1, 36, 32, "aspp d36"),
    (1, 1280, 256, 1, 1, 1, 32, "aspp project"),
    (1, 1024, 256, 3, 1, 1, 32, "aux 3x3 (fwd only)"),
]


def timeit(fn, reps):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3      # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--only", default="")
    ap.add_argument("--scale", type=int, default=1, help="2 = 512x512 inputs")
    ap.add_argument("--shapes", default="", help="comma-separated substrings of shape names to run")
    ap.add_argument("--opt", default="", help="name=value[,name=value] library options, e.g. conv_split=0")
    ap.add_argument("--camax", default="", help="weight gradient with per-channel maxima of: x, dy, or x,dy (as the BatchNorm kernels publish them)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    if args.opt:
        for kv in args.opt.split(","):
            k, v = kv.split("=")
            ops.set_option(k, v)
    B = args.batch
    tot = {"fwd": [0.0, 0.0], "dgrad": [0.0, 0.0], "wgrad": [0.0, 0.0]}
    print(f"{'shape':24s} {'pass':6s} {'us':>9s} {'nominal':>8s} {'executed':>9s} {'frac':>6s}  x count   kernel")
    want = [w for w in args.shapes.split(",") if w]
    for cnt, Cin, Cout, k, s, d, H, name in SHAPES:
        if want and not any(w in name for w in want):
            continue
        H = H * args.scale
        pad = (k // 2) * d if k > 1 else 0
        x = torch.randn(B, Cin, H, H, device=dev)
        w = torch.randn(Cout, Cin, k, k, device=dev) * 0.05
        wf, wd = ops.prep_weights(w)
        OH, OW = ops.conv_out_hw(H, H, k, s, pad, d)
        dy = torch.randn(B, Cout, OH, OW, device=dev)
        flops = 2.0 * B * OH * OW * Cout * Cin * k * k
        xc = x.abs().amax(dim=(0, 2, 3)).contiguous() if "x" in args.camax.split(",") else None
        dc = dy.abs().amax(dim=(0, 2, 3)).contiguous() if "dy" in args.camax.split(",") else None
        passes = {"fwd": lambda: ops.conv2d_fwd(x, wf, w.shape, s, pad, d),
                  "dgrad": lambda: ops.conv2d_dgrad(dy, wd, w.shape, x.shape, s, pad, d),
                  "wgrad": lambda: ops.conv2d_wgrad(x, dy, w.shape, s, pad, d, x_camax=xc, dy_camax=dc)}
        for pname, fn in passes.items():
            if args.only and pname not in args.only.split(","):
                continue
            if "fwd only" in name and pname != "fwd":
                continue
            if name == "conv1" and pname == "dgrad":
                continue
            us = timeit(fn, args.reps)
            tf = flops / us / 1e6
            # one instrumented launch: which kernel class ran, and how much of the nominal work it executed
            ops.prof_reset()
            ops.prof_enable(True)
            fn()
            torch.cuda.synchronize()
            ops.prof_enable(False)
            kname, work, exe = "?", flops, flops
            for c in range(ops.PROF_NCLASSES):
                n, _, w_, e_, _ = ops.prof_collect(c)
                if n and "conv" in ops.prof_class_name(c):
                    kname, work, exe = ops.prof_class_name(c), w_, e_
            ops.prof_reset()
            etf = tf * (exe / work if work else 1.0)
            split = "split" in kname
            nprod = (3.0 if ops.CONV_ARITH[0] == 1 else 6.0) if split else 1.0
            frac = etf * nprod / (PEAK16 if split else PEAK32)
            tot[pname][0] += us * cnt
            tot[pname][1] += flops * cnt
            tot[pname].append(exe / work * flops * cnt if work else flops * cnt)
            print(f"{name:24s} {pname:6s} {us:9.1f} {tf:8.1f} {etf:9.1f} {frac:6.3f}  x{cnt}   {kname}")
    for pname, v in tot.items():
        us, fl, ex = v[0], v[1], sum(v[2:])
        if us:
            print(f"TOTAL {pname:6s} {us / 1e3:8.2f} ms  {fl / us / 1e6:6.1f} TFLOP/s nominal, {ex / us / 1e6:6.1f} TFLOP/s executed (fp32-equivalent)")


if __name__ == "__main__":
    main()
