"""Several workgroups per channel in the resident BatchNorm kernels (round 5, "bn_coop"): results against one workgroup per
channel, bitwise reproducibility, and time per launch - forward and backward at the step's small-channel shapes."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from weaklysuperviseddl_amd import ops

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(2)


def timeit(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


for C, H in ((64, 64), (128, 32), (256, 32), (512, 32)):
    B = 16
    x = torch.randn(B, C, H, H, device=dev, generator=g)
    dy = torch.randn(B, C, H, H, device=dev, generator=g)
    gamma, beta = torch.rand(C, device=dev, generator=g) + 0.5, torch.randn(C, device=dev, generator=g) * 0.1
    res = {}
    for coop in (0, 256):
        ops.set_option("bn_coop", coop)

        def fwd():
            return ops.bn_train_fwd(x, gamma, beta, torch.zeros(C, device=dev), torch.ones(C, device=dev), 0.1, 1e-5, relu=True)

        y, mean, invstd = fwd()

        def bwd():
            return ops.bn_train_bwd(x, dy, None, gamma, mean, invstd, True, False, beta=beta)

        dx, dgam, dbet, _ = bwd()
        y2 = fwd()[0]
        dx2 = bwd()[0]
        torch.cuda.synchronize()
        res[coop] = (y.clone(), mean.clone(), invstd.clone(), dx.clone(), dgam.clone(), dbet.clone(), torch.equal(y, y2) and torch.equal(dx, dx2),
                     timeit(fwd), timeit(bwd))
    a, b = res[0], res[256]
    rel = lambda u, v: ((u - v).abs().max() / (v.abs().max() + 1e-30)).item()
    print(f"C={C:4d} {H}x{H}: fwd {a[7]:.1f} -> {b[7]:.1f} us, bwd {a[8]:.1f} -> {b[8]:.1f} us; max rel diff y {rel(b[0], a[0]):.1e} mean {rel(b[1], a[1]):.1e} "
          f"invstd {rel(b[2], a[2]):.1e} dx {rel(b[3], a[3]):.1e} dgamma {rel(b[4], a[4]):.1e} dbeta {rel(b[5], a[5]):.1e}; reproducible {a[6]} / {b[6]}; "
          f"finite {bool(torch.isfinite(b[0]).all() and torch.isfinite(b[3]).all())}")
ops.set_option("bn_coop", 0)
