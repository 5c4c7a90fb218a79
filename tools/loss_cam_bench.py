#!/usr/bin/env python3
"""Timing of the HBM-bound hot-path kernels at BASELINE config sizes: pairwise-affinity loss (cfg3 / cfg5),
LayerCAM epilogue (cfg1), cross-entropy, Adam.  Prints microseconds and GB/s against the algorithmic bytes of
SURVEY.md 8(d)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from weaklysuperviseddl_amd import ops  # noqa: E402
from weaklysuperviseddl_amd.optim import FlatAdam  # noqa: E402


def timeit(fn, reps=20):
    for _ in range(3):          # allocator growth at a new size costs milliseconds: not inside the timed calls
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


def main():
    dev = torch.device("cuda:0")
    print(f"{'kernel':44s} {'us':>9s} {'GB/s':>9s}  algorithmic bytes")
    for (B, H, W, name) in [(32, 256, 256, "cfg3"), (8, 512, 512, "cfg5")]:
        yy, xx = torch.meshgrid(torch.linspace(0, 1, H), torch.linspace(0, 1, W), indexing="ij")
        img = torch.stack([torch.sin(7 * xx + c) * torch.cos(5 * yy) for c in range(3)]).mul(0.5).add(0.5)
        img = img.unsqueeze(0).repeat(B, 1, 1, 1).to(dev).contiguous()
        preds = torch.randn(B, 2, H, W, device=dev)
        px = B * H * W
        pr = preds.clone().requires_grad_()
        us = timeit(lambda: ops.pairwise_affinity_loss(pr, img, 5, 0.1, 0.0, True, 0))
        print(f"{'ncut fwd+bwd fused ' + name:44s} {us:9.1f} {28 * px / us / 1e3:9.1f}  28 B/px (20 read + 8 write)")
        us = timeit(lambda: ops.pairwise_affinity_loss(preds, img, 5, 0.1, 0.0, True, 0))
        print(f"{'ncut fwd only ' + name:44s} {us:9.1f} {20 * px / us / 1e3:9.1f}  20 B/px")
        pb = torch.softmax(preds, 1).requires_grad_()
        us = timeit(lambda: ops.pairwise_affinity_loss(pb, img, 5, 0.1, 5.0, False, 1))
        print(f"{'boundary fwd+bwd fused ' + name:44s} {us:9.1f} {28 * px / us / 1e3:9.1f}  28 B/px")
        lab = torch.randint(0, 2, (B, H, W), device=dev)
        lg = preds.clone().requires_grad_()
        us = timeit(lambda: ops.cross_entropy(lg, lab))
        print(f"{'softmax-CE fwd+bwd ' + name:44s} {us:9.1f} {24 * px / us / 1e3:9.1f}  24 B/px (8 logits + 8 int64 label + 8 grad)")
        pl = torch.softmax(preds, 1).requires_grad_()
        us = timeit(lambda: ops.lovasz_softmax(pl, lab))
        print(f"{'lovasz-softmax fwd+bwd ' + name:44s} {us:9.1f} {24 * px / us / 1e3:9.1f}  24 B/px (8 probas + 8 int64 label + 8 grad; "
              "2 sorts of (key, index) pairs + 2 scans in between)")
    acts = [torch.relu(torch.randn(8, c, 14, 14, device=dev)) for c in (1024, 2048)]
    grads = [torch.randn(8, c, 14, 14, device=dev) * 1e-3 for c in (1024, 2048)]
    nbytes = sum(8 * a.numel() for a in acts) + 8 * 224 * 224 * 5
    us = timeit(lambda: ops.layercam_epilogue(acts, grads, (224, 224), 1.0, "modular", 0.3))
    print(f"{'layercam epilogue B=8 (3 launches)':44s} {us:9.1f} {nbytes / us / 1e3:9.1f}  {nbytes / 8 / 1e6:.2f} MB/img")
    n = 39_633_986
    p = [torch.randn(n, device=dev).requires_grad_()]
    opt = FlatAdam(p, lr=1e-4)
    opt.flat_grad.normal_()
    us = timeit(opt.step)
    print(f"{'adam 39.6M params':44s} {us:9.1f} {28 * n / us / 1e3:9.1f}  28 B/param")


if __name__ == "__main__":
    main()
