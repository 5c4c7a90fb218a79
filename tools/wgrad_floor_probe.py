"""Where does the weight gradient of the fp16x2 kernels lose accuracy when one element of x / dy is huge?"""
import os, sys
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from weaklysuperviseddl_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(78)
for (Cin, Cout, k, s, d, H, B) in [(1024, 256, 1, 1, 1, 16, 4), (1024, 256, 1, 1, 1, 32, 2), (256, 256, 3, 1, 2, 32, 4)]:
    pad = (k // 2) * d if k > 1 else 0
    for what in ("x", "dy", "both"):
        x = torch.randn(B, Cin, H, H, generator=g).to(dev)
        dy = torch.randn(B, Cout, H, H, generator=g).to(dev)
        if what in ("x", "both"):
            x[0, 3, H - 1, H - 1] = 2.0 ** 20
        if what in ("dy", "both"):
            dy[0, 5, 0, 0] = 2.0 ** 20
        ref = torch.nn.grad.conv2d_weight(x.double(), (Cout, Cin, k, k), dy.double(), s, pad, d)
        for opts in ({}, dict(wgrad_chan_scale=1), dict(wgrad_direct=0), dict(wgrad_dyraw=0), dict(conv_arith=0), dict(wgrad_split=0)):
            for o, v in opts.items():
                ops.set_option(o, v)
            dw = ops.conv2d_wgrad(x, dy, (Cout, Cin, k, k), s, pad, d)
            for o, v in opts.items():
                ops.set_option(o, {"wgrad_direct": 1, "wgrad_dyraw": 1, "conv_arith": 1, "wgrad_split": 1, "wgrad_chan_scale": 0}[o])
            e = (dw.double() - ref).abs()
            rel_el = (e / (ref.abs() + 1e-3)).max().item()
            row = (e.amax(dim=(1, 2, 3)) / ref.abs().amax(dim=(1, 2, 3)))
            col = (e.amax(dim=(0, 2, 3)) / ref.abs().amax(dim=(0, 2, 3)))
            print("Cin %d k %d H %d outlier in %-4s %-22s worst row %.1e (row %d) worst col %.1e (col %d) worst element %.1e" %
                  (Cin, k, H, what, opts or "default", row.max().item(), row.argmax().item(), col.max().item(), col.argmax().item(), rel_el), flush=True)
