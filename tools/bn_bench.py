#!/usr/bin/env python3
"""Timing of the train-mode BatchNorm kernels on the activation shapes of DeepLabV3-R50 at B=16, 256x256:
    python tools/bn_bench.py [--opt name=value,...]
forward (with / without residual + ReLU) and backward; microseconds and TB/s against the algorithmic bytes."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from weaklysuperviseddl_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--opt", default="")
args = ap.parse_args()
for kv in [x for x in args.opt.split(",") if x]:
    k, v = kv.split("=")
    ops.set_option(k, int(v))
dev = torch.device("cuda:0")


_blocker = torch.randn(8192, 8192, device=dev)


def timeit(fn, reps=30):
    """GPU time per call: the calls are enqueued behind a ~10 ms blocker so that the host's ~20 us per call (Python, ctypes,
    three torch.empty) is not what the events measure."""
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        torch.mm(_blocker, _blocker)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


tot = 0.0
print(f"{'C x HW (count / step)':26s} {'fwd us':>8s} {'TB/s':>6s} {'bwd us':>8s} {'TB/s':>6s}")
for C, HW, res, cnt in [(64, 4096, False, 7), (128, 1024, False, 8), (256, 1024, False, 19), (256, 4096, True, 3), (512, 1024, False, 11),
                        (1024, 1024, True, 7), (2048, 1024, True, 4)]:
    B = 16
    H = int(HW ** 0.5)
    x = torch.randn(B, C, H, H, device=dev)
    r = torch.randn(B, C, H, H, device=dev) if res else None
    g, b = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    dy = torch.randn(B, C, H, H, device=dev)
    y, mean, invstd, bits = ops.bn_train_fwd(x, g, b, rm, rv, 0.1, 1e-5, r, True, want_mask=True, mask_if=res)   # WSDL_BN_RELU_BITS=0: bits is None
    tf = timeit(lambda: ops.bn_train_fwd(x, g, b, rm, rv, 0.1, 1e-5, r, True, want_mask=True, mask_if=res))
    tb = timeit(lambda: ops.bn_train_bwd(x, dy, y if (res and bits is None) else None, g, mean, invstd, True, res,
                                         beta=None if res else b, relu_mask=bits))
    nb = x.numel() * 4
    bf, bb = nb * (3 if res else 2), nb * ((5 if bits is None else 4) if res else 3)     # algorithmic streams of 4 B / element
    tot += cnt * (tf + tb)
    print(f"{C:5d} x {HW:5d} (x{cnt:2d}) {'res' if res else '   '}     {tf:8.1f} {bf / tf / 1e6:6.2f} {tb:8.1f} {bb / tb / 1e6:6.2f}")
print(f"sum over the step's launches: {tot / 1e3:.3f} ms")
