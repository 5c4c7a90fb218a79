"""ASPP's four branch convolutions at B=16 (2048 -> 256, 32 x 32): one launch each against the grouped forward launch, and the
chain of four input-gradient launches against the multi-source one (round 5).  us per variant, HIP events, 20 repetitions."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from weaklysuperviseddl_amd import ops

dev = torch.device("cuda:0")
if "--opt" in sys.argv:                 # --opt name=value[,name=value]: library options (e.g. tile_img_major=0)
    i = sys.argv.index("--opt")
    for kv in sys.argv[i + 1].split(","):
        k, v = kv.split("=")
        ops.set_option(k, int(v))
    del sys.argv[i:i + 2]
B, C, Co, H = (int(a) for a in (sys.argv[1:5] if len(sys.argv) > 4 else (16, 2048, 256, 32)))
g = torch.Generator(device=dev).manual_seed(1)
ks, dils = [1, 3, 3, 3], [1, 12, 24, 36]
x = torch.randn(B, C, H, H, device=dev, generator=g)
ws = [torch.randn(Co, C, k, k, device=dev, generator=g) / (C * k * k) ** 0.5 for k in ks]
preps = [ops.prep_weights(w, True, True) for w in ws]
dys = [torch.randn(B, Co, H, H, device=dev, generator=g) for _ in ks]
xa = ops.amax_of(x)
das = [ops.amax_of(d) for d in dys]
xshape = tuple(x.shape)
shapes = [tuple(w.shape) for w in ws]
base = torch.zeros(xshape, device=dev)


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


def fwd_single():
    for (wf, _), w, d in zip(preps, ws, dils):
        ops.conv2d_fwd(x, wf, w.shape, 1, d * (w.shape[2] - 1) // 2, d, x_amax=xa)


def fwd_group():
    ops.conv2d_fwd_group(x, [p[0] for p in preps], shapes, dils, x_amax=xa)


def dgrad_chain():
    acc = base
    for dy, (_, wd), w, d, a in zip(dys, preps, ws, dils, das):
        acc = ops.conv2d_dgrad(dy, wd, w.shape, xshape, 1, d * (w.shape[2] - 1) // 2, d, accumulate_into=acc, dy_amax=a)


order = [1, 2, 3, 0]


def dgrad_multi():
    ops.conv2d_dgrad_multi([dys[i] for i in order], [preps[i][1] for i in order], [shapes[i] for i in order],
                           [dils[i] for i in order], xshape, accumulate_into=base, dy_amaxes=[das[i] for i in order])


for i, (w, d) in enumerate(zip(ws, dils)):
    wf, wd = preps[i]
    t = timeit(lambda: ops.conv2d_fwd(x, wf, w.shape, 1, d * (w.shape[2] - 1) // 2, d, x_amax=xa))
    t2 = timeit(lambda: ops.conv2d_dgrad(dys[i], wd, w.shape, xshape, 1, d * (w.shape[2] - 1) // 2, d, accumulate_into=base, dy_amax=das[i]))
    print(f"branch k={w.shape[2]} d={d}: fwd {t:.1f} us, dgrad (accumulating) {t2:.1f} us")
print(f"forward: four launches {timeit(fwd_single):.1f} us, grouped {timeit(fwd_group):.1f} us")
print(f"input gradient: chain of four {timeit(dgrad_chain):.1f} us, multi-source {timeit(dgrad_multi):.1f} us")
for tps in (20, 30, 45, 90):
    ops.set_option("group_tps10", tps)
    print(f"group_tps10 = {tps}: grouped forward {timeit(fwd_group):.1f} us")
ops.set_option("group_tps10", 20)
