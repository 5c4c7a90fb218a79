"""layer3's 256-channel 3x3 (128 tiles of 256 x 128: half the chip per K slice).  Would pairing a block's input
gradient with its weight gradient in ONE grid add anything to what the step already does - the two launched on two streams?
Times, for l3.conv2 (256 -> 256, 3x3, dilation 2, 16 x 32 x 32): each kernel alone, the two one after the other, the two on
two streams at once (as the training step issues them).   python tools/l3_pair_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from weaklysuperviseddl_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def timed(fn, reps=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for (name, Cin, Cout, k, d, H, B) in [("l3.conv2 d2", 256, 256, 3, 2, 32, 16), ("l4.conv2 d4", 512, 512, 3, 4, 32, 16)]:
    pad = d
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(B, Cin, H, H, device=dev, generator=g)
    dy = torch.randn(B, Cout, H, H, device=dev, generator=g)
    w = torch.randn(Cout, Cin, k, k, device=dev, generator=g) * 0.05
    wf, wd = ops.prep_weights(w)
    side = torch.cuda.Stream(device=dev)
    main = torch.cuda.current_stream(dev)

    def dgrad():
        ops.conv2d_dgrad(dy, wd, w.shape, x.shape, 1, pad, d)

    def wgrad():
        ops.conv2d_wgrad(x, dy, w.shape, 1, pad, d)

    def serial():
        dgrad()
        wgrad()

    def both():
        side.wait_stream(main)
        with torch.cuda.stream(side):
            wgrad()
        dgrad()
        main.wait_stream(side)

    t_d, t_w, t_s, t_b = timed(dgrad), timed(wgrad), timed(serial), timed(both)
    print(f"{name}: input gradient alone {t_d:.1f} us, weight gradient alone (with its pre-split and reduce) {t_w:.1f} us, one after the "
          f"other {t_s:.1f} us, on two streams at once {t_b:.1f} us ({100.0 * (t_s - t_b) / t_s:.0f} % of the serial time hidden)")
