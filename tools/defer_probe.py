#!/usr/bin/env python3
"""Probe for a schedule change (not built): would weight gradients of the last layers cost less beside the NEXT forward than
beside their own backward?  Adds ~2.6 ms of weight-gradient launches (the shapes of layer4 + ASPP, results discarded) on the
side stream of a normal training step, enqueued (a) nowhere, (b) at the start of the forward, (c) at the start of backward,
(d) both halves.  step(b) - step(a) against step(c) - step(a) is what moving that work would buy."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from weaklysuperviseddl_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
B, S = 16, 256
model, opt, _step, _ = bench.build_workload("cfg2", B, S, dev, 0)
img, masks = bench.synthetic_batch(B, S, S, dev, 1)
side = ops.side_stream(dev)
SHAPES = [((B, 512, 32, 32), (512, 512, 3, 3), 1, 4, 4)] * 3 + [((B, 512, 32, 32), (2048, 512, 1, 1), 1, 0, 1)] * 3 + \
         [((B, 2048, 32, 32), (512, 2048, 1, 1), 1, 0, 1)] * 2 + [((B, 2048, 32, 32), (256, 2048, 3, 3), 1, 12, 12),
                                                                  ((B, 2048, 32, 32), (256, 2048, 3, 3), 1, 24, 24)]
g = torch.Generator().manual_seed(0)
OPER = []
for xs, ws, st, pad, dil in SHAPES:
    x = torch.randn(xs, generator=g).to(dev)
    dy = torch.randn(xs[0], ws[0], xs[2], xs[3], generator=g).to(dev)
    out = torch.empty(ws, device=dev)
    OPER.append((x, dy, ws, st, pad, dil, out, ops.amax_of(x, True), ops.amax_of(dy, True)))


def dummy(part):
    cur = torch.cuda.current_stream(dev)
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        for x, dy, ws, st, pad, dil, out, xa, da in (OPER if part is None else OPER[part::2]):
            ops.conv2d_wgrad(x, dy, ws, st, pad, dil, out=out, x_amax=xa, dy_amax=da)


def step(mode):
    m = torch.clamp(masks, max=1)
    if mode in ("fwd", "both"):
        dummy(None if mode == "fwd" else 0)
    out = model(img)["out"]
    loss = ops.cross_entropy(out, m.long())
    opt.zero_grad()
    if mode in ("bwd", "both"):
        dummy(None if mode == "bwd" else 1)
    loss.backward()
    opt.step()
    return loss


def timeit(mode, steps=20, warm=5):
    for _ in range(warm):
        step(mode)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step(mode)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


# the dummy work alone
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    dummy(None)
torch.cuda.synchronize()
print(f"dummy weight gradients alone: {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms")
for rep in range(2):
    for mode in ("none", "fwd", "bwd", "both"):
        print(f"{mode:5s} {timeit(mode):7.3f} ms/step", flush=True)
