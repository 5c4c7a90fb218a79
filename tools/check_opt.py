#!/usr/bin/env python3
"""Bit-compare a library option against the default on a few conv shapes (fwd + dgrad): an option that only changes
scheduling must reproduce the default's output exactly.   python tools/check_opt.py wgrad_wide=1
(the option's value 0 is the baseline)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from weaklysuperviseddl_amd import ops
dev = torch.device("cuda:0")
opts = [kv.split("=") for kv in sys.argv[1].split(",")]
shapes = [(16, 512, 512, 3, 1, 4, 32), (16, 512, 2048, 1, 1, 1, 32), (16, 2048, 256, 3, 1, 12, 32), (16, 256, 256, 3, 1, 2, 32),
          (16, 256, 1024, 1, 1, 1, 32), (8, 512, 512, 3, 1, 4, 64), (32, 1024, 512, 1, 1, 1, 32), (16, 2048, 512, 1, 1, 1, 32),
          (2, 64, 64, 3, 1, 1, 64), (4, 128, 128, 3, 2, 1, 32), (2, 256, 64, 1, 1, 1, 16), (8, 2048, 256, 3, 1, 24, 28),
          (16, 64, 256, 1, 1, 1, 64), (16, 128, 128, 3, 1, 1, 32), (3, 48, 80, 3, 1, 1, 17), (16, 1024, 256, 3, 1, 1, 32),
          (2, 128, 128, 3, 1, 1, 8), (1, 256, 128, 3, 1, 2, 16), (3, 128, 256, 3, 1, 3, 24), (2, 256, 256, 3, 1, 36, 32),
          (2, 128, 128, 3, 1, 1, 28), (16, 2048, 256, 3, 1, 36, 32), (5, 128, 128, 5, 1, 1, 40)]
ok = True
for B, Cin, Cout, k, s, d, H in shapes:
    g = torch.Generator(device=dev).manual_seed(Cin + Cout)
    pad = (k // 2) * d if k > 1 else 0
    x = torch.randn(B, Cin, H, H, device=dev, generator=g)
    w = torch.randn(Cout, Cin, k, k, device=dev, generator=g) * 0.05
    res = {}
    for tag, vals in (("base", [(o, 0) for o, _ in opts]), ("opt", [(o, int(v)) for o, v in opts])):
        for o, v in vals:
            ops.set_option(o, v)
        wf, wd = ops.prep_weights(w)
        y = ops.conv2d_fwd(x, wf, w.shape, s, pad, d)
        dy = torch.randn(y.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
        dx = ops.conv2d_dgrad(dy, wd, w.shape, x.shape, s, pad, d)
        dw = ops.conv2d_wgrad(x, dy, w.shape, s, pad, d)
        res[tag] = (y, dx, dw)
    e = [bool(torch.equal(a, b)) for a, b in zip(res["base"], res["opt"])]
    md = [((a - b).abs().max() / a.abs().max()).item() for a, b in zip(res["base"], res["opt"])]
    print((B, Cin, Cout, k, s, d, H), "fwd/dgrad/wgrad bit-equal:", e, "max rel diff", md)
    ok &= all(e)
print("ALL EQUAL" if ok else "DIFFERENCES")
