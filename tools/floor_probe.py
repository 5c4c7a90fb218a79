"""Per-region accuracy of the split arithmetics on data with a huge dynamic range inside one tensor (the fp16x2 floor):
worst output channel / worst 8x8 block, max-norm error relative to the region's own maximum, against float64."""
import os, sys
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from weaklysuperviseddl_amd import ops
from test_hip_ops import _region_maxnorm_ratio, MODES
dev = torch.device("cuda:0")
MODES = dict(MODES)
MODES["fp16x2s"] = dict(conv_split=1, wgrad_split=1, conv_arith=2)
g = torch.Generator().manual_seed(78)
for data in ("unit", "outlier20", "outlier30", "outlier35", "outlier40", "graded30", "graded40"):
    for Cin, Cout, k, s, d, H, B in [(256, 256, 3, 1, 2, 32, 4), (1024, 256, 1, 1, 1, 16, 4)]:
        pad = (k // 2) * d if k > 1 else 0
        x = torch.randn(B, Cin, H, H, generator=g).to(dev)
        w = (torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5).to(dev)
        dy = torch.randn(B, Cout, H, H, generator=g).to(dev)
        if data.startswith("outlier"):
            x[0, 3, H - 1, H - 1] = 2.0 ** int(data[7:])
            dy[0, 5, 0, 0] = 2.0 ** int(data[7:])
        if data.startswith("graded"):
            e = int(data[6:])
            sc = torch.tensor([2.0 ** (-e * b / (B - 1)) for b in range(B)], device=dev).view(B, 1, 1, 1)
            x, dy = x * sc, dy * sc
        ref = F.conv2d(x.double(), w.double(), None, s, pad, d)
        ref_dx = torch.nn.grad.conv2d_input(x.shape, w.double(), dy.double(), s, pad, d)
        ref_dw = torch.nn.grad.conv2d_weight(x.double(), w.shape, dy.double(), s, pad, d)
        for mode in ("fp32", "bf16x3", "fp16x2", "fp16x2s"):
            for o, v in MODES[mode].items():
                ops.set_option(o, v)
            wf, wdg = ops.prep_weights(w)
            y = ops.conv2d_fwd(x, wf, w.shape, s, pad, d)
            dx = ops.conv2d_dgrad(dy, wdg, w.shape, x.shape, s, pad, d)
            dw = ops.conv2d_wgrad(x, dy, w.shape, s, pad, d)
            r_dw = ((dw.double() - ref_dw).abs().amax(dim=(1, 2, 3)) / ref_dw.abs().amax(dim=(1, 2, 3))).max().item()
            c_dw = ((dw.double() - ref_dw).abs().amax(dim=(0, 2, 3)) / ref_dw.abs().amax(dim=(0, 2, 3))).max().item()
            f, gd = _region_maxnorm_ratio(y, ref), _region_maxnorm_ratio(dx, ref_dx)
            print("%-10s Cin %4d k %d %-10s fwd ch %.1e blk %.1e | dgrad ch %.1e blk %.1e | wgrad worst cout row %.1e cin col %.1e" %
                  (data, Cin, k, mode, f[0], f[1], gd[0], gd[1], r_dw, c_dw), flush=True)
