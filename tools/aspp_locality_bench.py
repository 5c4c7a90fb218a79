"""ASPP grouped forward / multi-source input gradient: workgroup order against time (round 5).  python tools/aspp_locality_bench.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from weaklysuperviseddl_amd import ops

dev = torch.device("cuda:0")
B, C, Co, H = 16, 2048, 256, 32
g = torch.Generator(device=dev).manual_seed(1)
ks, dils = [1, 3, 3, 3], [1, 12, 24, 36]
x = torch.randn(B, C, H, H, device=dev, generator=g)
ws = [torch.randn(Co, C, k, k, device=dev, generator=g) / (C * k * k) ** 0.5 for k in ks]
preps = [ops.prep_weights(w, True, True) for w in ws]
dys = [torch.randn(B, Co, H, H, device=dev, generator=g) for _ in ks]
xa = ops.amax_of(x)
das = [ops.amax_of(d) for d in dys]
xshape, shapes = tuple(x.shape), [tuple(w.shape) for w in ws]
base = torch.zeros(xshape, device=dev)
order = [1, 2, 3, 0]


def timeit(fn, reps=20):
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


def fwd():
    ops.conv2d_fwd_group(x, [p[0] for p in preps], shapes, dils, x_amax=xa)


def dg():
    ops.conv2d_dgrad_multi([dys[i] for i in order], [preps[i][1] for i in order], [shapes[i] for i in order],
                           [dils[i] for i in order], xshape, accumulate_into=base, dy_amaxes=[das[i] for i in order])


ref = [t.clone() for t in ops.conv2d_fwd_group(x, [p[0] for p in preps], shapes, dils, x_amax=xa)]
for tps, il in ((30, 0), (20, 0), (20, 1)):
    ops.set_option("group_tps10", tps)
    ops.set_option("group_interleave", il)
    out = ops.conv2d_fwd_group(x, [p[0] for p in preps], shapes, dils, x_amax=xa)
    err = max(((a - b).abs().max() / b.abs().max()).item() for a, b in zip(out, ref))
    print(f"grouped forward, {tps / 10:.1f} taps per slice, interleave {il}: {timeit(fwd):.1f} us (max rel diff to the first form {err:.1e})")
ref = None
for rf, py in ((0, 0), (1, 1), (1, 2), (1, 4)):
    ops.set_option("ms_rowfast", rf)
    ops.set_option("ms_py", py)
    base.zero_()
    dg()
    cur = base.clone()
    if ref is None:
        ref = cur
    print(f"multi-source input gradient, rowfast {rf} py {py}: {timeit(dg):.1f} us (equal to the first form: {torch.equal(cur, ref)})")
ops.set_option("ms_rowfast", 1)
ops.set_option("ms_py", 0)
