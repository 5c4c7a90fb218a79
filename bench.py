#!/usr/bin/env python3
"""bench.py - SegmentationModel train img/s at B=16 256x256 (BASELINE.json metric, configs[1]).

One "step" = one pass of the hot path over one synthetic batch: DeepLabV3-ResNet50 forward (aux head
included, as the reference computes it), 2-class cross-entropy, backward, Adam - fp32, inputs resident
in HBM before the timed region.  N > 1: one process per GPU (torchrun), the batch of 16 is PER GPU
(weak scaling), gradients all-reduced over RCCL in 4 buckets overlapped with backward.

Prints ONE JSON line (rank 0).  Besides the contract keys:
  roofline     - the dominant kernel class (by summed device time): algorithmic FLOPs of its launches
                 divided by their summed HIP-event durations, against the fp32 MFMA peak (157.3 TFLOP/s).
                 Taken in a second, instrumented pass of the same K steps (event records around every
                 launch would perturb `value`); `kernels` lists every instrumented class.
  cpu_baseline - the CPU oracle (PyTorch CPU restatement, kind "port") timed on this box's host cores on a
                 bounded sample (B=4 of the same workload), rank 0, N=1 only.
  cam          - secondary metric of BASELINE.json ("CAM ms/img"): FrozenResNetCAM forward + class-logit
                 backward + LayerCAM epilogue + threshold on 8 x 224x224, batched.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
BF16_MFMA_PEAK_TFLOPS = 2516.6         # MI355X_MICROARCH.md: ~2.5 PF dense = 256 CUs x 4 SIMDs x 1024 FLOP/clk x 2.4 GHz
HBM_PEAK_GBS = 8000.0
# nominal dense FLOPs per image, DeepLabV3-R50 at 256x256, fwd (with aux) + bwd
GFLOP_PER_IMG_256 = 250.2


def synthetic_batch(B, H, W, device, seed):
    """rand image -> ImageNet normalise; blobby binary masks (9x9 box filter + threshold)."""
    g = torch.Generator().manual_seed(seed)
    img = torch.rand(B, 3, H, W, generator=g)
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    img = (img - mean) / std
    m = (torch.rand(B, 1, H, W, generator=g) > 0.5).float()
    m = torch.nn.functional.avg_pool2d(m, 9, 1, 4)
    masks = (m[:, 0] > 0.5).long()
    return img.to(device), masks.to(device)


def sync_all(world):
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def host_cores():
    """CPU share of this process: cgroup quota if there is one, else the affinity mask, capped at 16
    (the GPU box's share per GPU; torch would otherwise spawn one thread per host core)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 16))


def pmc_traffic(kernel):
    """HBM bytes per launch (fetch + write) of `kernel` from the committed rocprofv3 --pmc passes of this command
    (profiles/r*_pmc_traffic.json: FETCH_SIZE x 2 - gfx950 counts half of a coalesced read, calibrated on a
    known-size copy in our access widths - plus WRITE_SIZE).  PMC passes cannot run inside bench.py itself."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not files:
        return None
    k = json.load(open(files[-1])).get("kernels", {}).get(kernel)
    return None if not k else round(k["fetch_bytes_per_launch"] + k["write_bytes_per_launch"])


def cpu_baseline(B, H, W, steps=2):
    import oracle
    torch.manual_seed(0)
    threads = host_cores()
    torch.set_num_threads(threads)
    model = oracle.build_segmentation_model().train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    img, masks = synthetic_batch(B, H, W, "cpu", 1)

    def one():
        out = model(img)["out"]
        loss = torch.nn.functional.cross_entropy(out, torch.clamp(masks, max=1))
        opt.zero_grad()
        loss.backward()
        opt.step()

    log(f"cpu_baseline: oracle on {threads} threads, B={B}")
    one()                                  # warm-up
    log("cpu_baseline: warm-up step done")
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
        log("cpu_baseline: timed step done")
    dt = (time.perf_counter() - t0) / steps
    return {"value": round(B / dt, 4), "unit": "img/s", "cores": threads, "kind": "port",
            "sample": f"oracle SegmentationModel fwd+CE+bwd+Adam, B={B} {H}x{W}, 1 warm-up + {steps} timed steps, "
                      f"torch CPU {torch.__version__} on {threads} threads"}


def cam_bench(device, iters=5):
    from weaklysuperviseddl_amd.TraditionalModel import FrozenResNetCAM, LayerCAMGenerator
    torch.manual_seed(0)
    model = FrozenResNetCAM(37)
    g = torch.Generator().manual_seed(3)
    for m in model.modules():
        if hasattr(m, "running_mean"):
            m.running_mean.copy_(torch.randn(m.num_features, generator=g) * 0.1)
            m.running_var.copy_(torch.rand(m.num_features, generator=g) + 0.5)
    model = model.to(device).eval()
    gen = LayerCAMGenerator(model, ["layer3", "layer4"])
    imgs = torch.rand(8, 3, 224, 224, generator=g).to(device)
    cls = (torch.arange(8) % 37).to(device)
    for _ in range(2):
        gen.generate_batch(imgs, 1.0, cls, thresh=0.3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        gen.generate_batch(imgs, 1.0, cls, thresh=0.3)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / iters * 1e3
    return {"ms_per_img": round(ms / 8, 4), "batch": 8, "size": 224,
            "what": "FrozenResNetCAM fwd + class-logit bwd (to layer3 output) + LayerCAM epilogue + threshold"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16, help="images per GPU")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-cam", action="store_true")
    ap.add_argument("--cam-only", action="store_true", help="only the secondary CAM ms/img measurement (profiling aid)")
    ap.add_argument("--opt", default="", help="library options name=value[,name=value] (A/B experiments, e.g. ksplit_big=0)")
    ap.add_argument("--serial", action="store_true",
                    help="no side-stream overlap anywhere: every kernel has the chip to itself (profiling aid; the "
                         "roofline pass always runs like this)")
    args = ap.parse_args()

    from weaklysuperviseddl_amd import ops
    for kv in [x for x in args.opt.split(",") if x]:
        k, v = kv.split("=")
        ops.set_option(k, int(v))
    from weaklysuperviseddl_amd.dp import init_distributed, GradBucketReducer
    from weaklysuperviseddl_amd.TraditionalModel import build_segmentation_model, train_step
    from weaklysuperviseddl_amd.TraditionalModel.SegmentationModel import make_optimizer

    if args.cam_only:
        print(json.dumps({"cam": cam_bench(torch.device("cuda", 0), iters=10)}), flush=True)
        return
    if args.serial:
        ops.OVERLAP_WGRAD[0] = False
    rank, local, world = init_distributed()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torchrun --nproc-per-node {args.gpus}")
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    dev_index = local % torch.cuda.device_count()     # == local on a full node; rehearsal boxes have fewer GPUs
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)

    torch.manual_seed(0)                       # identical weights on every rank
    model = build_segmentation_model().to(device).train()
    opt = make_optimizer(model, lr=1e-4)
    reducer = GradBucketReducer(opt) if world > 1 else None
    B, S = args.batch, args.size
    img, masks = synthetic_batch(B, S, S, device, 1 + rank)

    def step():
        return train_step(model, opt, img, masks)

    if rank == 0:
        log(f"model on {device}, world={world}, B={B}, {S}x{S}; warm-up {args.warmup} steps")
    for i in range(args.warmup):
        step()
        if rank == 0 and i == 0:
            torch.cuda.synchronize()
            log("first step done")
    sync_all(world)
    if rank == 0:
        log(f"timing {args.steps} steps")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    sync_all(world)
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    ms_per_step = dt / args.steps * 1e3
    value = world * B * args.steps / dt
    if rank == 0:
        log(f"{ms_per_step:.2f} ms/step, {value:.1f} img/s")
    loss_val = float(loss.item())

    result = {
        "metric": "SegmentationModel train img/s at B=16 256x256",
        "value": round(value, 3), "unit": "img/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"DeepLabV3-ResNet50 (SegmentationModel, aux head computed) fwd + CrossEntropy + bwd + Adam, "
                               f"B={B}/GPU {S}x{S}x3, random-init weights, live dropout (BASELINE configs[1])",
                   "arithmetic": "fp32 tensors; conv products as 6 bf16 MFMAs on exact 3-way bf16 splits of both operands, "
                                 "fp32 accumulate (fp32-level accuracy, tools/conv_accuracy.py); stem / classifier convs on fp32 MFMA",
                   "global_batch": B * world, "image_size": S, "parallelism": f"dp{world}",
                   "final_loss": round(loss_val, 5)},
    }

    if not args.no_roofline:
        # instrumented pass: same steps on every rank (the collectives must match), HIP events around every
        # conv / loss launch on the launch stream on rank 0 only
        # The timed region overlaps weight-gradient kernels (side stream) with the main chain; a kernel's own
        # duration is only meaningful when it has the chip to itself, so this pass serialises the two streams.
        ops.OVERLAP_WGRAD[0] = False
        if rank == 0:
            ops.prof_reset()
            ops.prof_enable(True)
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        ops.prof_enable(False)
        ops.OVERLAP_WGRAD[0] = not args.serial
    if rank == 0 and not args.no_roofline:
        kernels = []
        for c in range(ops.PROF_NCLASSES):
            n, ms, work, exe, byt = ops.prof_collect(c)
            if n:
                kernels.append({"kernel": ops.prof_class_name(c), "launches": n, "avg_us": round(ms / n * 1e3, 3),
                                "total_ms": round(ms, 3), "work": work, "executed": exe, "alg_bytes": byt})
        ops.prof_reset()
        kernels.sort(key=lambda k: -k["total_ms"])
        if kernels:
            top = kernels[0]
            # `achieved` counts the MFMA work really issued (nominal dense FLOPs minus the K-chunks skipped because
            # every pixel of the tile reads zero padding under that tap); the nominal rate is given beside it
            ach = top["executed"] / (top["total_ms"] * 1e-3) / 1e12
            nom = top["work"] / (top["total_ms"] * 1e-3) / 1e12
            split = "split" in top["kernel"]
            if split:
                # bf16x3-split kernel: every fp32-equivalent FLOP is six bf16 MFMA FLOPs really issued; price those
                # against the dense bf16 MFMA peak (256 CUs x 4 SIMDs x 1024 FLOP/clk x 2.4 GHz)
                mfma, peak = 6.0 * ach, BF16_MFMA_PEAK_TFLOPS
                note = ("dense bf16 MFMA peak (v_mfma_f32_32x32x16_bf16); the kernel evaluates each fp32 product as six "
                        "bf16 partial products (operands split exactly into three bf16 pieces, fp32 accumulate), so "
                        "`achieved` = 6 x the fp32-equivalent rate `achieved_fp32_equivalent`")
            else:
                mfma, peak = ach, FP32_MFMA_PEAK_TFLOPS
                note = "fp32-input MFMA (v_mfma_f32_32x32x2_f32) dense peak"
            result["roofline"] = {"bound": "mfma", "kernel": top["kernel"], "achieved": round(mfma, 3),
                                  "peak": peak, "unit": "TFLOP/s", "frac": round(mfma / peak, 4),
                                  "achieved_fp32_equivalent": round(ach, 3), "achieved_nominal": round(nom, 3),
                                  "traffic": pmc_traffic(top["kernel"]),
                                  "algorithmic_bytes_per_launch": top["alg_bytes"] / top["launches"],
                                  "launches": top["launches"], "avg_launch_us": top["avg_us"],
                                  "flop_per_launch_avg": top["executed"] / top["launches"],
                                  "flop_per_launch_avg_nominal": top["work"] / top["launches"],
                                  "method": "second pass of the same steps, HIP events around every launch, wgrad side stream "
                                            "serialised (the timed region overlaps it with the main chain)",
                                  "peak_note": note}
            result["kernels"] = [{k: (round(v / 1e12, 3) if k in ("work", "executed") else (round(v / 1e9, 3) if k == "alg_bytes" else v))
                                  for k, v in kk.items()} |
                                 {"tflops": round(kk["executed"] / (kk["total_ms"] * 1e-3) / 1e12, 3),
                                  "tflops_nominal": round(kk["work"] / (kk["total_ms"] * 1e-3) / 1e12, 3)} for kk in kernels]
            result["model_tflops_nominal"] = round(value * GFLOP_PER_IMG_256 * (S / 256) ** 2 / 1e3, 3)

    if world > 1:
        dist.barrier()
    if rank == 0:
        if world == 1 and not args.no_cam:
            log("cam bench")
            result["cam"] = cam_bench(device)
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(4, S, S)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
