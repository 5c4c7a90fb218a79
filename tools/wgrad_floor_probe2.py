import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from weaklysuperviseddl_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(78)
Cin, Cout, k, H, B = 1024, 256, 1, 32, 2
for E in (20, 16, 12, 8):
    x = torch.randn(B, Cin, H, H, generator=g).to(dev)
    dy = torch.randn(B, Cout, H, H, generator=g).to(dev)
    x[0, 3, H - 1, H - 1] = 2.0 ** E
    dy[0, 5, 0, 0] = 2.0 ** E
    ref = torch.nn.grad.conv2d_weight(x.double(), (Cout, Cin, k, k), dy.double(), 1, 0, 1)[:, :, 0, 0]
    dw = ops.conv2d_wgrad(x, dy, (Cout, Cin, k, k), 1, 0, 1)[:, :, 0, 0].double()
    e = (dw - ref).abs()
    row = e.amax(1) / ref.abs().amax(1)
    r = row.argmax().item()
    c = e[r].argmax().item()
    print("E", E, "worst row", r, "%.2e" % row.max().item(), "at col", c, "ref", ref[r, c].item(), "got", dw[r, c].item(),
          "| dy at x-outlier pixel", dy[0, r, H - 1, H - 1].item(), "x at dy-outlier pixel", x[0, c, 0, 0].item())
    # error of column 3 elements relative to themselves, and of generic elements
    rel3 = (e[:, 3] / ref[:, 3].abs())
    print("   column 3: median rel err %.2e max %.2e ; generic elements: median rel err %.2e" %
          (rel3.median().item(), rel3.max().item(), (e / ref.abs().clamp(min=1.0)).median().item()))
