"""Soak: N training steps replayed from the launch plan against the same N steps issued eagerly - parameters, Adam moments and
BatchNorm buffers must be bit-identical at the end, with a learning-rate schedule and a changing batch in between.

    python tools/soak_plan.py [--steps 300] [--batch 8] [--size 128]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from weaklysuperviseddl_amd import plan  # noqa: E402
from weaklysuperviseddl_amd.TraditionalModel import build_segmentation_model, train_step  # noqa: E402
from weaklysuperviseddl_amd.TraditionalModel.SegmentationModel import make_optimizer  # noqa: E402


def run(planned, steps, B, S, dev):
    plan.PLAN_STEP[0] = planned
    torch.manual_seed(0)
    model = build_segmentation_model().to(dev).train()
    opt = make_optimizer(model, lr=1e-3)
    g = torch.Generator().manual_seed(7)
    batches = [(torch.randn(B, 3, S, S, generator=g).to(dev), ((torch.rand(B, S, S, generator=g) > 0.5).long() * 255).to(dev))
               for _ in range(5)]
    torch.manual_seed(99)
    losses = []
    t0 = time.perf_counter()
    for i in range(steps):
        opt.lr = 1e-3 * (0.5 ** (i // 50))                    # a step schedule: the plan reads lr from device memory
        img, m = batches[i % len(batches)]
        losses.append(train_step(model, opt, img, m))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = next(iter(opt.__dict__.get("_wsdl_planned", {}).values()), None)
    state = [opt.flat_param.clone(), opt.exp_avg.clone(), opt.exp_avg_sq.clone()] + [b.clone() for b in model.buffers()]
    return state, [float(l) for l in losses], dt, (st.records, st.replays) if st is not None else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--size", type=int, default=128)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    se, le, te, _ = run(False, a.steps, a.batch, a.size, dev)
    sp, lp, tp, counts = run(True, a.steps, a.batch, a.size, dev)
    same = sum(bool(torch.equal(x, y)) for x, y in zip(se, sp))
    print(f"{a.steps} steps B={a.batch} {a.size}x{a.size}: eager {te:.1f} s, planned {tp:.1f} s (records, replays) = {counts}")
    print(f"loss first / last: eager {le[0]:.6f} / {le[-1]:.6f}, planned {lp[0]:.6f} / {lp[-1]:.6f}; all losses equal: {le == lp}")
    print(f"state tensors bit-identical: {same} of {len(se)}; finite: {all(torch.isfinite(t.float()).all().item() for t in sp)}")
    sys.exit(0 if same == len(se) and le == lp else 1)


if __name__ == "__main__":
    main()
