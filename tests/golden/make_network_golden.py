#!/usr/bin/env python3
"""Whole-network gradient fixture at a BASELINE size: the fp64 run of the CPU oracle on
``bench.synthetic_batch(4, 256, 256)`` (B*H*W = 4096 samples per BatchNorm channel at the 32x32 maps, 16384 at
layer1: ReLU-mask flips average out, unlike the 64x64 case), dropout off, train-mode BatchNorm.

``--batch 16`` writes the same fixture at BASELINE cfg2's full per-GPU batch (network_grads_256_b16.npz): 16384 samples per
BatchNorm channel at the 32x32 maps and 16 in the ASPP image-pooling branch, so the fp32-vs-fp64 yardstick tightens.

Stored (tests/golden/network_grads_256.npz):
  loss64 / loss32, logits checksums and a 16-strided sample of the logits of both runs, per-parameter gradient norms of the fp64 run, the fp32 oracle's relative L2
  distance to fp64 for every parameter (the yardstick two correct fp32 implementations differ by), and the FULL fp64
  gradients (as float32) of backbone.conv1, layer1.0.conv1, layer4.2.conv3, classifier.4 and every BatchNorm
  weight / bias.  Weights are the seeded initialisation (torch.manual_seed(0)); `weights_checksum` guards that.

The two networks are third-party (torchvision) and unpinned by the reference: this fixture pins the HIP path against
the oracle's float64 arithmetic, not against reference outputs.  Run in the build container:
    python tests/golden/make_network_golden.py
"""
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import oracle  # noqa: E402

FULL = ["backbone.conv1.weight", "backbone.layer1.0.conv1.weight", "backbone.layer4.2.conv3.weight",
        "classifier.4.weight", "classifier.4.bias"]


def build(dtype):
    torch.manual_seed(0)
    m = oracle.build_segmentation_model()
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    return m.to(dtype).train()


def weights_checksum(model):
    return float(sum(p.detach().double().abs().sum().item() for p in model.parameters()))


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4, help="4: network_grads_256.npz; 16 (BASELINE cfg2's batch): network_grads_256_b16.npz")
    B = ap.parse_args().batch
    torch.set_num_threads(os.cpu_count() or 1)
    img, masks = bench.synthetic_batch(B, 256, 256, "cpu", 1)
    out = {}
    grads = {}
    for name, dtype in (("64", torch.float64), ("32", torch.float32)):
        t0 = time.time()
        m = build(dtype)
        logits = m(img.to(dtype))["out"]
        loss = F.cross_entropy(logits, masks)
        loss.backward()
        print(f"fp{name}: loss {loss.item():.9f}  ({time.time() - t0:.1f} s)", flush=True)
        out["loss" + name] = np.float64(loss.item())
        out["logits_abs_sum" + name] = np.float64(logits.detach().double().abs().sum().item())
        out["logits_sum" + name] = np.float64(logits.detach().double().sum().item())
        grads[name] = {k: p.grad.detach().double() for k, p in m.named_parameters() if p.grad is not None}
        out["logits_sample" + name] = logits.detach()[:, :, ::16, ::16].double().numpy()
        if name == "64":
            out["weights_checksum"] = np.float64(weights_checksum(m))
    names = sorted(grads["64"])
    out["names"] = np.array(names)
    out["norm64"] = np.array([grads["64"][k].norm().item() for k in names])
    out["maxabs64"] = np.array([grads["64"][k].abs().max().item() for k in names])
    out["err32_l2"] = np.array([((grads["32"][k] - grads["64"][k]).norm() / grads["64"][k].norm()).item() for k in names])
    out["err32_max"] = np.array([((grads["32"][k] - grads["64"][k]).abs().max() / grads["64"][k].abs().max()).item()
                                 for k in names])
    full = FULL if B == 4 else [k for k in FULL if k != "backbone.layer4.2.conv3.weight"] + ["backbone.layer3.0.conv1.weight"]
    for k in names:
        if k in full or ".bn" in k or k.endswith(".1.weight") or k.endswith(".1.bias") or "downsample.1" in k:
            if grads["64"][k].dim() == 1 or k in full:
                out["g:" + k] = grads["64"][k].numpy().astype(np.float32)
    out["batch"] = np.int64(B)
    path = os.path.join(ROOT, "tests", "golden", "network_grads_256.npz" if B == 4 else f"network_grads_256_b{B}.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB;", len(names), "parameters;",
          "median fp32-vs-fp64 rel L2 %.3e, max %.3e" % (np.median(out["err32_l2"]), out["err32_l2"].max()))


if __name__ == "__main__":
    main()
