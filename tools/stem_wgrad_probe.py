"""The stem's weight gradient (7x7, stride 2, 3 -> 64) at B = 16, 256 x 256: which kernel runs, how long, against float64."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from weaklysuperviseddl_amd import ops

dev = torch.device("cuda:0")
B, H = (int(sys.argv[1]) if len(sys.argv) > 1 else 16), (int(sys.argv[2]) if len(sys.argv) > 2 else 256)
x = torch.randn(B, 3, H, H, device=dev)
dy = torch.randn(B, 64, H // 2, H // 2, device=dev)
wshape = (64, 3, 7, 7)


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


ref = torch.nn.grad.conv2d_weight(x.double().cpu(), wshape, dy.double().cpu(), 2, 3, 1)
for opt in (1, 0):
    ops.set_option("stem_wgrad", opt)
    ops.launch_trace(True)
    ops.last_launches()
    dw = ops.conv2d_wgrad(x, dy, wshape, 2, 3, 1)
    tr = ops.last_launches()
    ops.launch_trace(False)
    err = ((dw.double().cpu() - ref).abs().max() / ref.abs().max()).item()
    print(f"stem_wgrad={opt}: {timeit(lambda: ops.conv2d_wgrad(x, dy, wshape, 2, 3, 1)):7.1f} us  max err vs float64 {err:.1e}   [{tr}]")
