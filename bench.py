#!/usr/bin/env python3
"""bench.py - SegmentationModel train img/s at B=16 256x256 (BASELINE.json metric, configs[1]).

One "step" = one pass of the hot path over one synthetic batch: DeepLabV3-ResNet50 forward (aux head
included, as the reference computes it), 2-class cross-entropy, backward, Adam - fp32, inputs resident
in HBM before the timed region.  N > 1: one process per GPU, the batch of 16 is PER GPU (weak scaling),
gradients all-reduced over RCCL in buckets that grow from the first layers (1 MB) to the last (48 MB), overlapped with
backward; each bucket's Adam launch and weight re-layout follow its collective on the side stream.

Launching: ``python bench.py --gpus N`` from a bare shell starts its own N ranks (child processes, before
this process touches the GPU) and prints rank 0's JSON line; under ``torchrun`` (WORLD_SIZE set) it is one
rank.  ``--config cfg2|cfg3|cfg4|cfg5`` selects the other BASELINE.json configurations (parity / rehearsal
cases; the default cfg2 is the metric's):
  cfg3  B=32/GPU 256x256, CE + 0.1 * LocalNormalizedCutLoss(sigma 0.1, w 5) on the logits
  cfg4  two-stage: LayerCAM -> pseudo masks on 16 x 224x224 per GPU -> in-memory hand-off (NEAREST 224->256) ->
        one training step on them; every step is the whole chain
  cfg5  B=8/GPU 512x512, CE + 0.1 * NCut + 0.1 * ConstrainToBoundaryLoss(sigma_c 0.1, sigma_s 5)

Prints ONE JSON line (rank 0).  Besides the contract keys:
  roofline     - the dominant kernel class (by summed device time): algorithmic FLOPs of its launches
                 divided by their summed HIP-event durations.  Taken in a second, instrumented pass of the same K
                 steps (event records around every launch would perturb `value`); `kernels` lists every class.
  cpu_baseline - the CPU oracle (PyTorch CPU restatement, kind "port") timed on this box's host cores on a
                 bounded sample of the same workload (cfg2: the full B=16 step, 1 warm-up + 3 timed), rank 0, N=1.
  cam          - secondary metric of BASELINE.json ("CAM ms/img"): FrozenResNetCAM forward + class-logit
                 backward + LayerCAM epilogue + threshold on 8 x 224x224, batched; with its own roofline and CPU
                 baselines (the reference's per-image B=1 loop form, and batched).
  ncut         - the pairwise-affinity loss kernel at cfg3 size beside the oracle's 24-slice formulation on the CPU.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

if (int(os.environ.get("WORLD_SIZE", "1")) > 1 or os.environ.get("WSDL_FORCE_DIST") == "1") and \
        os.environ.get("RANK", "0") == "0" and os.environ.get("NCCL_DEBUG", "").upper() not in ("INFO", "TRACE"):
    # Rank 0 of a multi-rank run asks RCCL to report its topology and its algorithm / protocol choices into a file that the
    # `dp.rccl` object of the JSON line is parsed from (rccl_debug_env below is the same rule for the ranks this script spawns
    # itself).  Set BEFORE torch is imported: the library reads its debug settings once.
    os.environ["NCCL_DEBUG"] = "INFO"
    os.environ["NCCL_DEBUG_SUBSYS"] = "INIT,GRAPH,TUNING"
    os.environ["NCCL_DEBUG_FILE"] = "/tmp/wsdl_rccl_rank0_%p.log"

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
BF16_MFMA_PEAK_TFLOPS = 2516.6         # MI355X_MICROARCH.md: ~2.5 PF dense = 256 CUs x 4 SIMDs x 1024 FLOP/clk x 2.4 GHz
HBM_PEAK_GBS = 8000.0
# nominal dense FLOPs per image, DeepLabV3-R50 at 256x256, fwd (with aux) + bwd
GFLOP_PER_IMG_256 = 250.2
# FrozenResNetCAM at 224x224: forward 12.4 GFLOP + the part of the class-logit backward LayerCAM consumes
# (fc, avgpool, layer4 down to layer3's output) ~5.9 GFLOP
CAM_GFLOP_PER_IMG = 18.3

CONFIGS = {
    "cfg2": dict(batch=16, size=256, what="fwd + CrossEntropy + bwd + Adam (BASELINE configs[1])"),
    "cfg3": dict(batch=32, size=256, what="fwd + CrossEntropy + 0.1*LocalNormalizedCutLoss(0.1, 5) + bwd + Adam (BASELINE configs[2])"),
    "cfg4": dict(batch=16, size=256, what="LayerCAM -> pseudo masks (16 x 224x224, thresh 0.3, keep_largest) -> in-memory "
                                          "hand-off -> fwd + CrossEntropy + bwd + Adam on them (BASELINE configs[3])"),
    "cfg5": dict(batch=8, size=512, what="fwd + CrossEntropy + 0.1*NCut + 0.1*ConstrainToBoundaryLoss(0.1, 5) + bwd + Adam "
                                         "(BASELINE configs[4])"),
}


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


def smooth_images(B, H, W, seed):
    """Piece-wise smooth RGB in [0,1] (the pairwise losses carry no signal on iid-uniform pixels)."""
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.linspace(0, 1, H), torch.linspace(0, 1, W), indexing="ij")
    f = torch.rand(B, 3, 4, 3, generator=g) * torch.tensor([3.0, 3.0, 6.28])
    img = torch.zeros(B, 3, H, W)
    for k in range(4):
        img += 0.25 * torch.sin(6.28 * (f[:, :, k, 0, None, None] * yy + f[:, :, k, 1, None, None] * xx) + f[:, :, k, 2, None, None])
    return (img * 0.5 + 0.5 + 0.01 * torch.randn(B, 3, H, W, generator=g)).clamp(0, 1)


class stdout_to_stderr:
    """Process-group creation prints from C++ ("[Gloo] Rank 0 is connected to ...") on file descriptor 1; stdout carries
    the one JSON line and nothing else."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)


def sync_all(world):
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def host_cores_available():
    """CPU share of this process: the affinity mask, cut by the cgroup quota if there is one (no cap of ours)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


CPU_THREAD_CAP = int(os.environ.get("WSDL_CPU_THREADS", "16"))   # this pool gives a one-GPU job 16 host cores


def host_cores():
    """Threads the CPU baselines run on: the cores this process may use (affinity mask / cgroup quota), capped at the
    pool's per-GPU CPU share (16; WSDL_CPU_THREADS changes it).  The bench line reports the threads used (``cores``), the
    uncapped count (``cores_available``) and the cap."""
    return max(1, min(host_cores_available(), CPU_THREAD_CAP))


def cpu_info():
    return {"cores": host_cores(), "cores_available": host_cores_available(), "thread_cap": CPU_THREAD_CAP,
            "cpu": cpu_model()}


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _pmc_file():
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not files:
        return None, {}
    d = json.load(open(files[-1]))
    return files[-1], d


def kernel_matches(cls, name):
    """Is the profiler's kernel `name` an instantiation of the library's timing class `cls` (wsdl_prof_class_name)?  Class names
    carry placeholders where instantiations differ (AR = the arithmetic, BK / MODE / DYRAW = any) and may leave trailing template
    arguments out (those must then be the profiler's printed defaults: anything)."""
    from weaklysuperviseddl_amd import ops
    name = name.replace("(anonymous namespace)::", "").replace("wsdl::", "")
    if name.startswith("void "):
        name = name[5:]
    name = name.split("(")[0].strip()
    base = cls.split("<")[0]
    if name.split("<")[0] != base:
        return False
    want = [t.strip() for t in cls[len(base):].strip("<>").split(",")] if "<" in cls else []
    got = [t.strip() for t in name[len(base):].strip("<>").split(",")] if "<" in name else []
    if len(want) > len(got):
        return False
    for w, g in zip(want, got):
        if w == "AR":
            if g != str(ops.CONV_ARITH[0] if ops.CONV_ARITH[0] != 1 else 1):
                return False
        elif w in ("BK", "MODE", "DYRAW", "MF"):
            continue
        elif w != g:
            return False
    return True


def pmc_traffic(kernel, with_source=False):
    """HBM bytes per launch (fetch + write) of `kernel` from the committed rocprofv3 --pmc passes of this command
    (profiles/r*_pmc_traffic.json: FETCH_SIZE x 2 - gfx950 counts half of a coalesced read, calibrated on a
    known-size copy in our access widths - plus WRITE_SIZE).  PMC passes cannot run inside bench.py itself: the figure
    is a LOOKUP, and `with_source` returns (bytes, "file (collected date)") so that the line says where it comes from."""
    path, d = _pmc_file()
    if path is None:
        return (None, None) if with_source else None
    kernels = d.get("kernels", {})
    cands = [v for kk, v in kernels.items() if kernel_matches(kernel, kk)]
    k = max(cands, key=lambda v: v.get("launches", 0)) if cands else None
    val = None if not k else round(k["fetch_bytes_per_launch"] + k["write_bytes_per_launch"])
    if with_source:
        src = f"{os.path.relpath(path, ROOT)} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, collected {d.get('collected', 'in an earlier round')})"
        return val, src
    return val


def mfma_busy_3x3():
    """MFMA-busy fraction of the 3x3 convolutions (north_star: ">= 40 % MFMA util on the 3x3 convs") from the committed
    per-shape counter runs, profiles/r*_mfma_busy_3x3.txt (tools/mfma_busy_3x3.sh: one rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES
    GRBM_GUI_ACTIVE run per SURVEY 8a shape and pass) - a lookup, like `traffic`.  Since round 6 every (shape, pass) is weighted
    by its LAUNCHES PER STEP and a pass's denominator holds its satellite launches (slab / split-K reduces, dY pre-split): the
    pass keys are that figure, `<pass>_matrix_kernels_alone` the matrix kernels' own."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_mfma_busy_3x3.txt")))
    if not files:
        return None
    out = {"source": os.path.relpath(files[-1], ROOT)}
    for line in open(files[-1]):
        if line.startswith("3x3 convolutions"):
            parts = line.split()
            out[parts[2]] = float(parts[-1])
            m = re.search(r"matrix kernels alone\s+([0-9.]+)", line)
            if m:
                out[parts[2] + "_matrix_kernels_alone"] = float(m.group(1))
                out["weighting"] = "launches per step; satellite launches in the denominator"
    return out if len(out) > 1 else None


def rocprof_avg_us(kernel):
    """Average duration of `kernel` in the committed ``rocprofv3 --kernel-trace --stats`` run of this command with the streams
    serialised (profiles/r*_bench_n1_serial_kernel_stats.csv) - the figure the HIP-event average of the run must agree
    with.  -> (us, calls, file) or (None, None, None)."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_n1_serial_kernel_stats.csv")))
    if not files:
        return None, None, None
    best = None
    for row in csv.DictReader(open(files[-1])):
        if kernel_matches(kernel, row["Name"]) and (best is None or int(row["Calls"]) > int(best["Calls"])):
            best = row
    if best is None:
        return None, None, os.path.relpath(files[-1], ROOT)
    return round(float(best["AverageNs"]) / 1e3, 3), int(best["Calls"]), os.path.relpath(files[-1], ROOT)


# The 256 x 128 forward / input-gradient kernel has three ENTRY POINTS around one body (conv_split.h: conv_igemm_split_body<256, 128, 4, 16,
# 512, AR>): the plain launch, the grouped launch of several convolutions (ASPP's forward) and the multi-source launch (ASPP's
# input gradient).  Each has a timing class of its own (so that a class lines up with one row of `rocprofv3 --stats`); the
# roofline line prices the BODY - all three together, the same set of convolutions the single class of rounds 1-4 held - and
# lists the entry points beside it.
BODY_256x128 = ("conv_igemm_split_kernel<256, 128, 4, BK, 512, AR, false, false>",
                "conv_igemm_split_group_kernel<256, 128, 4, 16, 512, AR, false>",
                "conv_igemm_split_kernel<256, 128, 4, 16, 512, AR, false, true>")
BODY_256x128_NAME = "conv_igemm_split_body<256, 128, 4, 16, 512, AR> (entry points: conv_igemm_split_kernel<256, 128, 4, 16, 512, AR, false, false | true>, conv_igemm_split_group_kernel<256, 128, 4, 16, 512, AR, false>)"


def merge_body_classes(kernels):
    """kernels: the per-class records of the instrumented pass -> the same list with the three entry points of the 256 x 128 body
    merged into one record (its parts kept under "entries")."""
    parts = [k for k in kernels if k["kernel"] in BODY_256x128]
    if len(parts) < 2:
        return kernels
    tot = {"kernel": BODY_256x128_NAME, "launches": sum(k["launches"] for k in parts),
           "total_ms": round(sum(k["total_ms"] for k in parts), 3), "work": sum(k["work"] for k in parts),
           "executed": sum(k["executed"] for k in parts), "alg_bytes": sum(k["alg_bytes"] for k in parts), "entries": parts}
    tot["avg_us"] = round(tot["total_ms"] / tot["launches"] * 1e3, 3)
    return [tot] + [k for k in kernels if k["kernel"] not in BODY_256x128]


def rocprof_avg_us_many(kernels):
    """rocprof_avg_us over several classes: (total duration / total calls, calls, file, per-class rows)."""
    rows, f = [], None
    for k in kernels:
        us, calls, f = rocprof_avg_us(k)
        if us is not None:
            rows.append({"kernel": k, "avg_us": us, "calls": calls})
    if not rows:
        return None, None, f, rows
    calls = sum(r["calls"] for r in rows)
    return round(sum(r["avg_us"] * r["calls"] for r in rows) / calls, 3), calls, f, rows


# ------------------------------------------------------------------------------------------ CPU baselines (oracle)
def _timed(fn, warm, steps, what):
    for _ in range(warm):
        fn()
    log(f"cpu_baseline: {what}: warm-up done")
    ts = []
    for _ in range(steps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
        log(f"cpu_baseline: {what}: timed iteration {len(ts)}/{steps} {ts[-1]:.2f} s")
    ts.sort()
    return ts[len(ts) // 2]                                   # median


def cpu_baseline(B, H, W, extra=None, steps=3):
    """BASELINE.md section 4 item 1 (item 4 at cfg5 size): the oracle's training step on the host cores."""
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
        if extra is not None:
            loss = loss + extra(out, img)
        opt.zero_grad()
        loss.backward()
        opt.step()

    dt = _timed(one, 1, steps, f"oracle train step B={B} {H}x{W} on {threads} threads")
    return {"value": round(B / dt, 4), "unit": "img/s", **cpu_info(), "kind": "port",
            "sample": f"oracle SegmentationModel fwd+loss+bwd+Adam, B={B} {H}x{W} (the full batch of the workload), 1 warm-up + "
                      f"{steps} timed steps, median; torch CPU {torch.__version__} on {threads} threads"}


def cam_cpu_baselines(n_img=8):
    """oracle FrozenResNetCAM + LayerCAM + threshold + keep_largest on 8 x 224x224, in the
    reference's per-image B=1 loop form (PsuedoMasks.py:47-65) and batched."""
    import oracle
    torch.manual_seed(0)
    threads = host_cores()
    torch.set_num_threads(threads)
    model = oracle.FrozenResNetCAM(37)
    g = torch.Generator().manual_seed(3)
    for m in model.modules():
        if hasattr(m, "running_mean"):
            m.running_mean.copy_(torch.randn(m.num_features, generator=g) * 0.1)
            m.running_var.copy_(torch.rand(m.num_features, generator=g) + 0.5)
    model.eval()
    gen = oracle.LayerCAMGenerator(model, ["layer3", "layer4"])
    imgs = torch.rand(n_img, 3, 224, 224, generator=g)
    cls = torch.arange(n_img) % 37

    def loop():
        for i in range(n_img):
            cam = gen.generate(imgs[i], alpha=1.0, class_idx=cls[i:i + 1])
            oracle.keep_largest(oracle.cam_to_mask(cam[0], 0.3))

    def batched():
        cam = gen.generate(imgs, alpha=1.0, class_idx=cls)
        for i in range(n_img):
            oracle.keep_largest(oracle.cam_to_mask(cam[i], 0.3))

    t_loop = _timed(loop, 1, 3, "oracle CAM per-image loop")
    t_bat = _timed(batched, 1, 3, "oracle CAM batched")
    base = {"unit": "ms/img", **cpu_info(), "kind": "port"}
    return {"per_image_loop": dict(base, value=round(t_loop / n_img * 1e3, 3),
                                   sample=f"oracle FrozenResNetCAM fwd + class-logit bwd (to the image, as the reference) + LayerCAM "
                                          f"+ threshold + keep_largest, {n_img} x 224x224 one image per call, 1 warm-up + 3 timed, median"),
            "batched": dict(base, value=round(t_bat / n_img * 1e3, 3),
                            sample=f"same, one batch of {n_img}")}


def ncut_cpu_baseline(B, H, W, steps=2):
    """LocalNormalizedCutLoss fwd+bwd in the reference's 24-slice formulation."""
    import oracle
    threads = host_cores()
    torch.set_num_threads(threads)
    img = smooth_images(B, H, W, 5)
    preds = torch.randn(B, 2, H, W, generator=torch.Generator().manual_seed(6)).requires_grad_()
    crit = oracle.LocalNormalizedCutLoss(0.1, 5)

    def one():
        preds.grad = None
        crit(preds, img).backward()

    dt = _timed(one, 1, steps, f"oracle NCut fwd+bwd ({B},2,{H},{W})")
    return {"value": round(dt * 1e3, 2), "unit": "ms/step", **cpu_info(), "kind": "port",
            "sample": f"oracle LocalNormalizedCutLoss(0.1, 5) fwd+bwd, preds ({B},2,{H},{W}), 24-slice formulation "
                      f"(AlternatingDirectionCutLoss.py:87-101), 1 warm-up + {steps} timed, median"}


# ------------------------------------------------------------------------------------------ secondary GPU legs
def cam_setup(device, n_img=8):
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
    imgs = torch.rand(n_img, 3, 224, 224, generator=g).to(device)
    cls = (torch.arange(n_img) % 37).to(device)
    return gen, imgs, cls


def conv_class_totals(ops):
    """(ms, nominal flop, executed flop, split-kernel executed flop) summed over the instrumented conv classes."""
    ms = work = exe = exe_split = 0.0
    for c in range(ops.PROF_NCLASSES):
        n, t, w, e, _ = ops.prof_collect(c)
        name = ops.prof_class_name(c)
        if n and "conv" in name:
            ms, work, exe = ms + t, work + w, exe + e
            if "split" in name:
                exe_split += e
    return ms, work, exe, exe_split


def cam_bench(device, iters=int(os.environ.get("WSDL_CAM_ITERS", "20")), roofline=True):
    from weaklysuperviseddl_amd import ops
    gen, imgs, cls = cam_setup(device)
    n_img = imgs.shape[0]
    # warm-up: eager pass, capture, and enough replays for the clocks to settle after the idle stretch of the set-up (with 2
    # warm-up calls the first measurement of a process read anything from 0.34 to 0.53 ms/img)
    for _ in range(12):
        gen.generate_batch(imgs, 1.0, cls, thresh=0.3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        gen.generate_batch(imgs, 1.0, cls, thresh=0.3)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / iters * 1e3
    out = {"ms_per_img": round(ms / n_img, 4), "batch": n_img, "size": 224,
           "launch": "hipGraph replay of the batch (LayerCAMGenerator.generate_batch, WSDL_CAM_SELF_GRAPH=0: eager)",
           "what": "FrozenResNetCAM fwd + class-logit bwd (to layer3 output) + LayerCAM epilogue + threshold"}
    # stage 1 with several batches in flight, both ways generate_pseudo_masks can run it: device_batch = 0 (its default: one
    # launch sequence per loader batch, three of them in flight - the masks do not depend on how batches are merged) and
    # device_batch = 32 (throughput option: the loader's batches of 8 merged into device batches of 32 images)
    nb, lanes = int(os.environ.get("WSDL_CAM_NB", "12")), int(os.environ.get("WSDL_CAM_LANES", "3"))
    db = int(os.environ.get("WSDL_CAM_DEVICE_BATCH", "32"))

    def in_flight(dbatch):
        for _ in range(3):
            gen.generate_coalesced([imgs] * nb, 1.0, [cls] * nb, 0.3, streams=lanes, device_batch=dbatch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            gen.generate_coalesced([imgs] * nb, 1.0, [cls] * nb, 0.3, streams=lanes, device_batch=dbatch)
        torch.cuda.synchronize()
        return round((time.perf_counter() - t0) / iters * 1e3 / (nb * n_img), 4)

    out["ms_per_img_in_flight"] = in_flight(0)
    out["in_flight"] = (f"{nb} loader batches of {n_img}, {lanes} in flight, one launch sequence each "
                        "(generate_pseudo_masks' default, device_batch = 0)")
    out["ms_per_img_pipelined"] = in_flight(db) if db > 0 else out["ms_per_img_in_flight"]
    out["pipelined"] = (f"{nb} loader batches of {n_img} merged into device batches of {db}, {lanes} in flight "
                        f"(LayerCAMGenerator.generate_coalesced; generate_pseudo_masks(device_batch={db}))" if db > 0 else out["in_flight"])
    if roofline:
        ops.prof_reset()
        ops.prof_enable(True)
        for _ in range(iters):
            gen._generate_batch_eager(imgs, 1.0, cls, 0.3)      # eager launches: a hipGraph replay records no per-launch events
        torch.cuda.synchronize()
        ops.prof_enable(False)
        kms, work, exe, exe_split = conv_class_totals(ops)
        ops.prof_reset()
        # whole-leg figure: needed-only FLOPs of the leg / its wall time; conv-kernel figure: executed FLOPs / event time
        leg_tf = CAM_GFLOP_PER_IMG * n_img / ms                        # GFLOP / ms = TFLOP/s (fp32-equivalent)
        conv_tf = exe / (kms * 1e-3) / 1e12 if kms else 0.0
        nprod = 3.0 if ops.CONV_ARITH[0] == 1 else 6.0
        out["roofline"] = {"bound": "mfma", "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                           "achieved": round(leg_tf, 3), "frac": round(leg_tf / BF16_MFMA_PEAK_TFLOPS, 4),
                           "achieved_mfma_issue": round(nprod * leg_tf, 3), "frac_mfma_issue": round(nprod * leg_tf / BF16_MFMA_PEAK_TFLOPS, 4),
                           "conv_kernels_fp32_equivalent": round(conv_tf, 3),
                           "conv_kernels_frac": round(nprod * conv_tf / BF16_MFMA_PEAK_TFLOPS, 4),
                           "mfma_products_per_fp32_product": nprod,
                           "conv_kernel_ms_per_batch": round(kms / iters, 4),
                           "gflop_per_img": CAM_GFLOP_PER_IMG,
                           "traffic": pmc_traffic("conv_igemm_split_kernel<128, 64, 2, 32, 256, AR>"),
                           "traffic_layercam_partial_kernel": pmc_traffic("layercam_partial_kernel"),
                           "traffic_source": pmc_traffic("layercam_partial_kernel", with_source=True)[1],
                           "note": "achieved / frac: needed-only fp32-equivalent FLOPs (forward 12.4 + backward to layer3's output 5.9 "
                                   "GFLOP/img) / wall time of the whole leg, against the dense 16-bit MFMA peak (*_mfma_issue: x the 16-bit "
                                   "MFMA products per fp32 product); "
                                   "conv_kernels_* = executed FLOPs of the instrumented conv launches / their HIP-event time"}
    return out


def ncut_bench(device, B=32, H=256, W=256, reps=20):
    from weaklysuperviseddl_amd import ops
    img = smooth_images(B, H, W, 5).to(device)
    preds = torch.randn(B, 2, H, W, generator=torch.Generator().manual_seed(6)).to(device).requires_grad_()
    for _ in range(3):
        ops.pairwise_affinity_loss(preds, img, 5, 0.1, 0.0, True, 0)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        ops.pairwise_affinity_loss(preds, img, 5, 0.1, 0.0, True, 0)      # forward AND gradient: one fused launch
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) / reps * 1e3
    px = B * H * W
    return {"us_fwd_bwd": round(us, 2), "shape": [B, 2, H, W],
            "roofline": {"bound": "hbm", "achieved": round(28 * px / us / 1e3, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(28 * px / us / 1e3 / HBM_PEAK_GBS, 4), "traffic": pmc_traffic("pairwise_kernel"),
                         "traffic_source": pmc_traffic("pairwise_kernel", with_source=True)[1],
                         "note": "28 B/px algorithmic (20 read + 8 gradient write, C=2), fused forward + backward launch; not bound by "
                                 "HBM nor by its exponentials (the cached-affinity form takes 88 us, staging the tile + halo is "
                                 "more than half: profiles/r03_notes.md)"}}


# ------------------------------------------------------------------------------------------ launching
def rank_environments(n, ndev, base_env, port=None):
    """The environment of each of the ``n`` ranks ``python bench.py --gpus n`` starts on a box with ``ndev`` GPUs (pure:
    nothing is launched - also the GPU-less dry run of the spawn logic, tests/test_abi_and_host.py).  One rank per GPU
    over RCCL when there are enough GPUs; with fewer, a rehearsal: ranks share the devices (at most 6 processes per
    GPU on this pool) and the collectives go over gloo."""
    import socket
    env = dict(base_env)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in env:
        if port is None:
            with socket.socket() as s:
                s.bind(("127.0.0.1", 0))
                port = s.getsockname()[1]
        env["MASTER_PORT"] = str(port)
    env["WORLD_SIZE"] = env["LOCAL_WORLD_SIZE"] = str(n)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, host_cores() // n)))
    if ndev < n:
        if n > 6 * max(ndev, 1):
            raise SystemExit(f"--gpus {n}: only {ndev} GPU(s) here and at most 6 processes may share one")
        env.setdefault("WSDL_DIST_BACKEND", "gloo")
    return [rccl_debug_env(dict(env, RANK=str(r), LOCAL_RANK=str(r)), r) for r in range(n)]


def spawn_ranks(args, argv):
    """``python bench.py --gpus N`` from a bare shell: start N ranks as child processes BEFORE anything here touches
    the GPU (no re-exec of a process that has initialised HIP), relay rank 0's JSON line, fail if any rank fails."""
    n = args.gpus
    ndev = torch.cuda.device_count()              # does not initialise the GPU on this image
    envs = rank_environments(n, ndev, os.environ)
    if ndev < n:
        log(f"{ndev} GPU(s) for {n} ranks: rehearsal over gloo, ranks share the device(s)")
    procs = []
    for r, e in enumerate(envs):
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=e,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    rcs = wait_ranks(procs, timeout_s=float(os.environ.get("WSDL_BENCH_TIMEOUT", "3000")))
    out0 = procs[0].captured
    for line in out0.splitlines():                # the contract: ONE JSON line on stdout; anything else a rank printed -> stderr
        (sys.stdout if line.startswith("{") else sys.stderr).write(line + "\n")
    sys.stdout.flush()
    if any(rcs):
        raise SystemExit(f"bench ranks exited with {rcs}")


def wait_ranks(procs, timeout_s=3000.0, poll_s=0.2):
    """Wait for every rank; rank 0's stdout is drained by a thread (``procs[0].captured``).  The FIRST rank that exits
    non-zero - or the deadline - ends the run: the others would otherwise sit in their collectives until the backend's
    own timeout (a rank that died before its first all-reduce leaves its peers waiting for it).  Children are killed by
    their exact PIDs."""
    import threading
    buf = []
    reader = threading.Thread(target=lambda: buf.append(procs[0].stdout.read()), daemon=True)
    if procs[0].stdout is not None:
        reader.start()
    deadline = time.monotonic() + timeout_s
    failed = None
    while True:
        rcs = [p.poll() for p in procs]
        if all(rc is not None for rc in rcs):
            break
        bad = [i for i, rc in enumerate(rcs) if rc not in (None, 0)]
        if bad or time.monotonic() > deadline:
            failed = bad[0] if bad else -1
            log(f"rank {failed} exited with {rcs[failed]}: stopping the other ranks" if bad else
                f"ranks still running after {timeout_s:.0f} s: stopping them")
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            t_kill = time.monotonic() + 10.0
            while any(p.poll() is None for p in procs) and time.monotonic() < t_kill:
                time.sleep(poll_s)
            for p in procs:
                if p.poll() is None:
                    p.kill()
            break
        time.sleep(poll_s)
    rcs = [p.wait() for p in procs]
    if procs[0].stdout is not None:
        reader.join(timeout=10.0)
    procs[0].captured = (buf[0] if buf else b"").decode()
    if failed == -1:
        rcs = [rc if rc else -9 for rc in rcs]
    return rcs


# ------------------------------------------------------------------------------------------ self-description of a data-parallel run
RCCL_ALGOS = {0: "Tree", 1: "Ring", 2: "CollnetDirect", 3: "CollnetChain", 4: "NVLS", 5: "NVLSTree"}
RCCL_PROTOS = {0: "LL", 1: "LL128", 2: "Simple"}


def rccl_debug_env(env, rank, log_dir="/tmp"):
    """Rank 0 of a multi-rank run asks RCCL to say what it does (topology at init, algorithm / protocol per collective size)
    into a file bench.py parses afterwards (``dp.rccl``); the caller's own NCCL_DEBUG settings win."""
    if int(rank) != 0 or env.get("NCCL_DEBUG", "").upper() in ("INFO", "TRACE"):
        return env              # (a caller who asked for INFO / TRACE output himself keeps it; VERSION / WARN are raised to INFO)
    env = dict(env)
    env["NCCL_DEBUG"] = "INFO"
    env["NCCL_DEBUG_SUBSYS"] = "INIT,GRAPH,TUNING"
    env["NCCL_DEBUG_FILE"] = os.path.join(log_dir, "wsdl_rccl_rank0_%p.log")
    return env


def parse_rccl_log(text, keep=24):
    """What RCCL reported: per (bytes, algorithm, protocol) how many collectives took it (NCCL_DEBUG_SUBSYS=TUNING lines
    "<Coll>: <n> Bytes -> Algo <a> proto <p> time <t>"), the topology lines of the communicator's set-up, and whether every
    collective ran on rings only ("verify it is not ring-only")."""
    import re
    choices, topo, version = {}, [], None
    pat = re.compile(r"(\w+): (\d+) Bytes -> Algo (\d+) proto (\d+)(?: time ([0-9.eE+-]+))?")
    for line in text.splitlines():
        msg = line.split("NCCL INFO", 1)[1].strip() if "NCCL INFO" in line else line.strip()
        m = pat.search(msg)
        if m:
            key = (m.group(1), int(m.group(2)), int(m.group(3)), int(m.group(4)))
            ent = choices.setdefault(key, {"collective": key[0], "bytes": key[1], "algo": RCCL_ALGOS.get(key[2], str(key[2])),
                                           "proto": RCCL_PROTOS.get(key[3], str(key[3])), "calls": 0, "model_time_us": None})
            ent["calls"] += 1
            if m.group(5):
                ent["model_time_us"] = float(m.group(5))
            continue
        if re.search(r"(RCCL|NCCL) version", msg) and version is None:
            version = msg
        if re.search(r"^(Channel|Ring|Trees|Connected all|\d+ coll channels|comm 0x|Using network|P2P|Setting affinity)", msg) \
                and len(topo) < keep:
            if msg.startswith("Channel") and sum(t.startswith("Channel") for t in topo) >= 2:
                continue                # (one line per channel, up to 128 of them: the first two say what they look like)
            topo.append(msg[:200])
    chosen = sorted(choices.values(), key=lambda e: -e["bytes"])
    algos = sorted({e["algo"] for e in chosen})
    return {"version_line": version, "choices": chosen[:keep], "algorithms_used": algos,
            "ring_only": (algos == ["Ring"]) if algos else None, "topology_lines": topo}


def dp_self_description(world, backend, rccl_version, per_rank_ms, bucket_ms, n1_reference, rccl_log_text):
    """The ``dp`` fields that let the first run on a real node be read without the builder present."""
    return {"rccl": {"nranks": int(world), "backend": backend, "rccl_version": rccl_version,
                     "debug": parse_rccl_log(rccl_log_text) if rccl_log_text else None,
                     "debug_note": "rank 0 ran with NCCL_DEBUG=INFO NCCL_DEBUG_SUBSYS=INIT,GRAPH,TUNING into a file; `choices` = the "
                                   "(algorithm, protocol) RCCL picked per collective size, `ring_only` answers SURVEY section 5"},
            "ms_per_step_by_rank": [None if v is None else round(float(v), 3) for v in per_rank_ms],
            "bucket_allreduce_ms": bucket_ms,
            "bucket_allreduce_ms_is": "per bucket (launch order): side-stream time from the point the bucket's all-reduce was enqueued to "
                                      "its completion on that stream (HIP events, 5 steps after the timed region, mean) - queueing behind "
                                      "earlier collectives included",
            "n1_reference": n1_reference}


def build_workload(cfg, B, S, device, rank, graph=False):
    """-> (step callable returning the loss tensor, description)."""
    from weaklysuperviseddl_amd import ops
    from weaklysuperviseddl_amd.TraditionalModel import (build_segmentation_model, train_step, LocalNormalizedCutLoss,
                                                         ConstrainToBoundaryLossSingle)
    from weaklysuperviseddl_amd.TraditionalModel.SegmentationModel import make_optimizer
    torch.manual_seed(0)                       # identical weights on every rank
    model = build_segmentation_model().to(device).train()
    opt = make_optimizer(model, lr=1e-4)
    extra = None
    if cfg in ("cfg3", "cfg5"):
        ncut = LocalNormalizedCutLoss(0.1, 5)
        if cfg == "cfg3":
            # (ops.scale_mean(x, w) = w * x.mean() as a launch of the library, so that the step's launch plan sees it)
            extra = lambda o, i: ops.scale_mean(ncut(o, i), 0.1)                                         # noqa: E731
        else:
            bnd = ConstrainToBoundaryLossSingle(0.1, 5, 5)
            def extra(o, i):
                o1, o2 = ops.fanout(o, 2)       # (two consumers: gradients summed by the library, not by autograd's own add)
                return ops.add_scalars(ops.scale_mean(ncut(o1, i), 0.1), ops.scale_mean(bnd(ops.softmax_channels(o2), i), 0.1))
    if cfg == "cfg4":
        from weaklysuperviseddl_amd.TraditionalModel import generate_pseudo_masks, stage_handoff
        gen, _, _ = cam_setup(device, 1)
        g = torch.Generator().manual_seed(100 + rank)
        imgs224 = torch.rand(B, 3, 224, 224, generator=g)               # stage-1 inputs, un-normalised
        labels = (torch.arange(B) + rank) % 37
        loader = [(imgs224.to(device), (labels, None))]

        def step():
            generate_pseudo_masks(loader, gen, cam_thresh=0.3, keep_largest_masks=True, write_png=False, device=device,
                                  keep_on_device=True)
            masks = generate_pseudo_masks.last_masks
            img, m = stage_handoff(loader[0][0], masks, (S, S), device)
            return train_step(model, opt, img, m.long())
    else:
        if cfg == "cfg2":
            img, masks = synthetic_batch(B, S, S, device, 1 + rank)
        else:
            # the pairwise losses need piece-wise smooth images
            _, masks = synthetic_batch(B, S, S, device, 1 + rank)
            mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
            std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
            img = ((smooth_images(B, S, S, 11 + rank) - mean) / std).to(device)

        def eager_step():
            return train_step(model, opt, img, masks, extra)
        step = eager_step
        if graph:
            from weaklysuperviseddl_amd.graph import GraphedTrainStep
            gstep = GraphedTrainStep(model, opt, extra, warmup=2)

            def step():
                return gstep(img, masks)
        return model, opt, step, eager_step
    return model, opt, step, step


def _plan_pending(opt):
    """Is the training step still going to record a launch plan (plan.PlannedTrainStep: neither recorded nor refused yet)?"""
    from weaklysuperviseddl_amd import plan as _plan
    if not _plan.PLAN_STEP[0]:
        return False
    table = opt.__dict__.get("_wsdl_planned", {})
    if not table:
        return True
    pst = next(iter(table.values()))
    red = getattr(opt, "_wsdl_reducer", None)
    if red is not None and not _plan.PLAN_DP[0]:
        return False
    return pst.plan is None and pst.disabled is None


def _plan_on(opt):
    pst = next(iter(opt.__dict__.get("_wsdl_planned", {}).values()), None)
    return pst is not None and pst.plan is not None and pst.replays > 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=None, help="images per GPU (default: the config's)")
    ap.add_argument("--size", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-cam", action="store_true")
    ap.add_argument("--no-full-width", action="store_true", help="skip the 5-step bf16x3 / fp32-MFMA legs of the default run")
    ap.add_argument("--cam-only", action="store_true", help="only the secondary CAM ms/img measurement (profiling aid)")
    ap.add_argument("--opt", default="", help="library options name=value[,name=value] (A/B experiments, e.g. ksplit_big=0)")
    ap.add_argument("--serial", action="store_true",
                    help="no side-stream overlap anywhere: every kernel has the chip to itself (profiling aid; the "
                         "roofline pass always runs like this)")
    ap.add_argument("--graph", type=int, default=None, help="1/0: force hipGraph replay of the training step on/off")
    ap.add_argument("--plan", type=int, default=None, help="0: every step eager (no launch-plan replay; the A/B partner)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args, sys.argv[1:])
        return

    from weaklysuperviseddl_amd import ops
    for kv in [x for x in args.opt.split(",") if x]:
        k, v = kv.split("=")
        ops.set_option(k, int(v))
    from weaklysuperviseddl_amd.dp import init_distributed, GradBucketReducer

    if args.plan is not None:
        from weaklysuperviseddl_amd import plan as _p
        _p.PLAN_STEP[0] = bool(args.plan)
    if args.cam_only:
        print(json.dumps({"cam": cam_bench(torch.device("cuda", 0), iters=10)}), flush=True)
        return
    if args.serial:
        ops.OVERLAP_WGRAD[0] = False
    if os.environ.get("WSDL_WGRAD_AFTER_DGRAD"):
        ops.WGRAD_AFTER_DGRAD[0] = True             # A/B: enqueue the input gradient before the weight gradient
    if os.environ.get("WSDL_DUMP_AFTER"):            # debugging aid: where is every thread N seconds from now?
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["WSDL_DUMP_AFTER"]), repeat=False, file=sys.stderr)
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 or os.environ.get("WSDL_FORCE_DIST") == "1":
        # (under torchrun the driver's environment arrives here unchanged: rank 0 turns RCCL's own report on before the init)
        os.environ.update(rccl_debug_env(dict(os.environ), os.environ.get("RANK", "0")))
    with stdout_to_stderr():
        rank, local, world = init_distributed()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    dev_index = local % torch.cuda.device_count()     # == local on a full node; rehearsal boxes have fewer GPUs
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)

    cfg = args.config
    B = args.batch or CONFIGS[cfg]["batch"]
    S = args.size or CONFIGS[cfg]["size"]
    use_graph = bool(args.graph) if args.graph is not None else False
    if use_graph and (world > 1 or dist.is_initialized() or cfg == "cfg4"):
        raise SystemExit("--graph 1: hipGraph replay covers the single-process training step (cfg2 / cfg3 / cfg5)")
    model, opt, step, eager_step = build_workload(cfg, B, S, device, rank, graph=use_graph)
    # WSDL_FORCE_DIST=1 puts the data-parallel machinery (RCCL broadcasts / bucketed all-reduces, control exchange) on a
    # single rank as well: the one-GPU rehearsal of the code path the 8-GPU run takes
    dp_on = world > 1 or (dist.is_available() and dist.is_initialized())
    n1_reference = None
    if dp_on and cfg != "cfg4":
        # the same process layout WITHOUT data parallelism: every rank steps its own replica on its own GPU, no collective
        # (the reducer, built right after, broadcasts rank 0's state over whatever these steps did) - the N = 1 figure this
        # run's scaling should be read against, measured on this node, in these processes
        for _ in range(max(args.warmup, 3)):
            step()
        sync_all(world)
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        sync_all(world)
        d1 = (time.perf_counter() - t1) / args.steps
        n1_reference = {"ms_per_step": round(d1 * 1e3, 3), "img_s_per_gpu": round(B / d1, 2), "steps": args.steps,
                        "what": "every rank stepping its own replica with no gradient exchange (same processes, same node, all GPUs "
                                "busy at once): what N x this figure would be is perfect scaling"}
        opt.__dict__.pop("_wsdl_planned", None)          # (plans recorded without the reducer's hooks are not this run's)
    with stdout_to_stderr():
        reducer = GradBucketReducer(opt, modules=[model]) if dp_on else None   # noqa: F841  (hooks live on the optimizer)

    if rank == 0:
        log(f"{cfg} on {device}, world={world}, B={B}, {S}x{S}; warm-up {args.warmup} steps")
    for i in range(args.warmup):
        step()
        if rank == 0 and i == 0:
            torch.cuda.synchronize()
            log("first step done")
    # The step records its launch plan on its third eligible call (under data parallelism: once the reducer has settled) - a call
    # that does the work of three steps.  It belongs to the warm-up: with fewer warm-up steps than that, a few more untimed
    # ones are taken until the plan stands (or has been refused); the timed region is then K replays, as in a long run.
    warmup_extra = 0
    while warmup_extra < 8 and not use_graph and _plan_pending(opt):
        step()
        warmup_extra += 1
    sync_all(world)
    if rank == 0:
        log(f"timing {args.steps} steps" + (f" (after {warmup_extra} more warm-up steps: launch plan recorded)" if warmup_extra else ""))
    cpu0 = time.process_time()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    host_issue_s = time.perf_counter() - t0           # the host has enqueued the K steps
    sync_all(world)
    dt = time.perf_counter() - t0
    cpu_s = time.process_time() - cpu0
    dt_local = dt
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    ms_per_step = dt / args.steps * 1e3
    value = world * B * args.steps / dt
    if rank == 0:
        log(f"{ms_per_step:.2f} ms/step, {value:.1f} img/s")
    loss_val = float(loss.item())

    metric = ("SegmentationModel train img/s at B=16 256x256" if cfg == "cfg2"
              else f"SegmentationModel train img/s, {cfg} (B={B} {S}x{S})")
    result = {
        "metric": metric,
        "value": round(value, 3), "unit": "img/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "warmup_extra": warmup_extra,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{cfg}: DeepLabV3-ResNet50 (SegmentationModel, aux head computed) {CONFIGS[cfg]['what']}, "
                               f"B={B}/GPU {S}x{S}x3, random-init weights, live dropout",
                   "arithmetic": ("fp32 tensors; conv products as 3 fp16 MFMAs on exact 2-way fp16 splits of both operands (22 "
                                  "significant bits; power-of-two scales per tensor in the forward / input gradient, per CHANNEL in the weight gradients), fp32 accumulate"
                                  if ops.CONV_ARITH[0] == 1 else
                                  "fp32 tensors; conv products as 6 bf16 MFMAs on exact 3-way bf16 splits of both operands, "
                                  "fp32 accumulate") + " - rms error vs fp64 at the level of the exact-fp32 MFMA chain "
                                 "(tools/conv_accuracy.py); stem / classifier convs on fp32 MFMA",
                   "global_batch": B * world, "image_size": S, "parallelism": f"dp{world}",
                   "backend": (dist.get_backend() if dp_on else None),
                   "launch": ("hipGraph replay (one host call per step)" if use_graph else
                              "launch-plan replay (one host call per step; see host.launch_plan)" if _plan_on(opt) else
                              "eager (one host call per kernel)"),
                   "final_loss": round(loss_val, 5)},
        "host": {"cpu_s_per_step": round(cpu_s / args.steps, 5),
                 "enqueue_wall_ms_per_step": round(host_issue_s / args.steps * 1e3, 3)},
    }
    # What the host needs to ISSUE one step: a step call into an EMPTY queue (synchronise, time the call alone), median of 7.
    # The wall time until K back-to-back steps are enqueued (enqueue_wall_ms_per_step, the figure of earlier rounds) also
    # contains the runtime's back-pressure: once the host is a few steps ahead, launches block until the GPU has drained
    # the queue, so a host that issues a step in 2.5 ms reads ~9 ms there (tools/plan_probe.py).
    singles = []
    for _ in range(7):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        step()
        singles.append(time.perf_counter() - t1)
    sync_all(world)
    singles.sort()
    result["host"]["issue_ms_per_step"] = round(singles[len(singles) // 2] * 1e3, 3)
    result["host"]["note"] = ("issue_ms_per_step: host time of one step call issued into an empty queue (median of 7, after the timed "
                              "region); enqueue_wall_ms_per_step: wall time until the K timed steps were enqueued / K - includes the "
                              "runtime's back-pressure once the host runs ahead of the GPU; cpu_s_per_step: process CPU time of all "
                              "threads (the runtime's helper threads included)")
    from weaklysuperviseddl_amd import plan as _plan
    pst = next(iter(opt.__dict__.get("_wsdl_planned", {}).values()), None) if not use_graph else None
    result["host"]["launch_plan"] = (None if pst is None else
                                     {"replays": pst.replays, "records": pst.records, "disabled": pst.disabled,
                                      "ops": None if pst.plan is None else pst.plan.stats,
                                      "verified": pst.plan is not None,
                                      "what": "steps after the second are ONE host call: the launches of an eager step recorded behind "
                                              "the C ABI (wsdl_plan_*), verified bit for bit against an eager step on a probe batch, "
                                              "then replayed from a C loop (weaklysuperviseddl_amd/plan.py)"})
    # optimiser tail on the main stream (join with the weight-gradient / collective stream + Adam), in a short pass of its
    # own: two event records per step are kept out of the timed region
    opt.time_tail = True
    for _ in range(5):
        step()
    opt.time_tail = False
    tail = opt.tail_ms()
    issue_by_rank = [result["host"]["issue_ms_per_step"]]
    if dp_on and world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, issue_by_rank[0])
        issue_by_rank = gathered
    result["host"]["issue_ms_per_step_by_rank"] = issue_by_rank
    result["optimizer_tail_ms"] = None if tail is None else round(tail, 4)
    torch.cuda.synchronize()
    result["range"] = dict(ops.range_status(device),
                           note="range sentinel of the fp16x2 arithmetic (one power-of-two scale per tensor): the BatchNorm kernels publish "
                                "max|tensor| and the smallest non-zero channel maximum; worst_log2 = the largest log2 of "
                                "their ratio over the last step's tensors, exceeded = some tensor beyond 2^25 (there conv_arith = 2, the "
                                "range guard of the forward / input gradient, is the arithmetic to use: WSDL_RANGE_GUARD=auto, the default, "
                                "switches it on for the steps that follow; the weight gradients scale every channel by its own power of two "
                                "from step 0 - the maxima come from the BatchNorm kernels)",
                           guard=__import__("weaklysuperviseddl_amd.optim", fromlist=["x"]).RANGE_GUARD[0],
                           guard_active=bool(__import__("weaklysuperviseddl_amd.optim", fromlist=["x"]).RANGE_GUARD_ACTIVE[0]))
    result["streams"] = ops.stream_census(device)
    if dp_on and dist.get_backend() != "nccl":
        result["config"]["rehearsal"] = ("ranks share the GPU(s) and the gradients travel over gloo through host memory: a rehearsal "
                                         "of the data-parallel code path, not a performance figure")
    if dp_on:
        per_rank = [dt_local / args.steps * 1e3]
        if world > 1:
            per_rank = [None] * world
            dist.all_gather_object(per_rank, dt_local / args.steps * 1e3)
        reducer.time_buckets = True
        for _ in range(5):
            step()
        reducer.time_buckets = False
        bucket_ms = reducer.bucket_times_ms()
        log_text = None
        if rank == 0 and os.environ.get("NCCL_DEBUG_FILE"):
            path = os.environ["NCCL_DEBUG_FILE"].replace("%p", str(os.getpid())).replace("%h", os.uname().nodename)
            try:
                log_text = open(path, errors="replace").read()
            except OSError:
                log_text = None
        try:
            rccl_version = ".".join(str(v) for v in torch.cuda.nccl.version()) if dist.get_backend() == "nccl" else None
        except Exception:
            rccl_version = None
        plan = reducer.comm_budget()
        result["dp"] = dp_self_description(world, dist.get_backend(), rccl_version, per_rank, bucket_ms, n1_reference, log_text)
        result["dp"] |= {"buckets": len(reducer.bucket_size), "early_launches_last_step": reducer.last_early_launches,
                        "control_exchanges": {"blocking": reducer.control_exchanges_blocking,
                                              "asynchronous_one_step_behind": reducer.control_exchanges_async},
                        "exposed_ms": None if tail is None else round(tail, 4),
                        "exposed_ms_is": "main-stream time per step from the last backward kernel to the parameters being written "
                                         "(5 steps after the timed region): the wait for the gradient collectives that did not overlap "
                                         "backward + the last Adam segment.  The same region of a plain single-process run is its "
                                         "`optimizer_tail_ms` (join with the weight-gradient stream + one Adam launch, ~0.3 ms): the "
                                         "difference is the exposed communication",
                        "bucket_plan": plan,
                        "exposed_bucket": plan[0] if plan else None,
                        "note": "bucket_plan: bytes per all-reduce (launch order: last bucket first, from gradient-ready hooks) and "
                                "the time each needs on xGMI (~153 GB/s per link, one link to each peer): ring_us through one "
                                "link, direct_us = reduce-scatter + all-gather over all links at once.  Only bucket 0 (the stem + "
                                "layer1, the last gradients of backward) cannot overlap backward: its time is the expected "
                                "exposed communication per step; the others have the rest of backward (>= 1 ms each) to hide in"}

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
            if world == 1:
                torch.cuda.synchronize()  # no hand-over from the previous step's side-stream work inside the first kernel's bracket
            eager_step()                  # HIP events around the launches: eager, never the graph replay
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
        per_class = sorted(kernels, key=lambda k: -k["total_ms"])
        kernels = sorted(merge_body_classes(kernels), key=lambda k: -k["total_ms"])
        if kernels:
            top = kernels[0]
            # `achieved` counts the MFMA work really issued (nominal dense FLOPs minus the K-chunks skipped because
            # every pixel of the tile reads zero padding under that tap); the nominal rate is given beside it
            ach = top["executed"] / (top["total_ms"] * 1e-3) / 1e12
            nom = top["work"] / (top["total_ms"] * 1e-3) / 1e12
            split = "split" in top["kernel"]
            if split:
                # split kernel: every fp32-equivalent FLOP is NPROD 16-bit MFMA FLOPs really issued (3 for the default
                # fp16x2 arithmetic, 6 for bf16x3); price those against the dense 16-bit MFMA peak
                # (256 CUs x 4 SIMDs x 1024 FLOP/clk x 2.4 GHz; fp16 and bf16 MFMAs run at the same rate)
                nprod = 3.0 if ops.CONV_ARITH[0] == 1 else 6.0
                mfma, peak = nprod * ach, BF16_MFMA_PEAK_TFLOPS
                note = ("dense 16-bit MFMA peak (v_mfma_f32_32x32x16_f16 / _bf16); the kernel evaluates each fp32 product as "
                        f"{int(nprod)} 16-bit partial products (operands split exactly into 16-bit pieces, fp32 accumulate), so "
                        f"`achieved` = {int(nprod)} x the fp32-equivalent rate `achieved_fp32_equivalent`; the fp32-equivalent "
                        f"ceiling of the scheme is peak / {int(nprod)} = {BF16_MFMA_PEAK_TFLOPS / nprod:.0f} TFLOP/s")
            else:
                mfma, peak = ach, FP32_MFMA_PEAK_TFLOPS
                note = "fp32-input MFMA (v_mfma_f32_32x32x2_f32) dense peak"
            entries = top.get("entries")
            if entries:
                rp_us, rp_calls, rp_file, rp_rows = rocprof_avg_us_many([e["kernel"] for e in entries])
                traffic_kernel = max(entries, key=lambda e: e["total_ms"])["kernel"]
            else:
                (rp_us, rp_calls, rp_file), rp_rows = rocprof_avg_us(top["kernel"]), None
                traffic_kernel = top["kernel"]
            step_tf = value / world * GFLOP_PER_IMG_256 * (S / 256) ** 2 / 1e3          # per GPU, nominal dense fp32-equivalent
            result["roofline"] = {"bound": "mfma", "kernel": top["kernel"],
                                  # ALGORITHMIC flops per launch (2 * P * Cout * K dense fp32-equivalent, padding taps
                                  # counted) / the kernel's average launch duration (HIP events of this run) / the peak of the pipe
                                  # the kernel runs on
                                  "achieved": round(nom, 3), "peak": peak, "unit": "TFLOP/s", "frac": round(nom / peak, 4),
                                  "frac_is": ("SURVEY 8(d)'s fraction: nominal algorithmic fp32-equivalent FLOPs of the launches of this "
                                              "kernel / their HIP-event time / the dense peak of the matrix pipe it runs on"
                                              + (" (16-bit MFMA: each fp32 product is evaluated as 3 fp16 MFMAs on exact 2-way splits, so "
                                                 "the scheme's own ceiling is peak / 3; against the fp32-input MFMA peak of 157.3 TFLOP/s "
                                                 "the same rate would read > 1).  frac_mfma_issue is the rate of 16-bit MFMA work really "
                                                 "ISSUED (3 per product, padding-only taps skipped) over the same peak" if split else "")),
                                  "frac_algorithmic_executed": round(ach / peak, 4),
                                  "frac_mfma_issue": round(mfma / peak, 4), "achieved_mfma_issue": round(mfma, 3),
                                  "frac_of_scheme_ceiling": round(ach / (peak / (nprod if split else 1)), 4),
                                  "step_frac": round(step_tf / peak, 4), "step_tflops_nominal": round(step_tf, 3),
                                  "step_frac_is": f"whole step: {GFLOP_PER_IMG_256 * (S / 256) ** 2 * B / 1e3:.3f} TFLOP nominal (fwd + bwd, "
                                                  "BASELINE.md section 3) / ms_per_step / the same peak",
                                  "static_fields": ["traffic", "traffic_source", "mfma_busy_3x3", "avg_launch_us_rocprof",
                                                    "rocprof_calls", "rocprof_source"],
                                  "static_note": "static_fields are LOOKUPS in committed rocprofv3 runs under profiles/ (counter passes and "
                                                 "the profiler cannot run inside bench.py); everything else in this object is measured in "
                                                 "this run",
                                  "achieved_fp32_equivalent_executed": round(ach, 3),
                                  "mfma_products_per_fp32_product": (nprod if split else 1),
                                  "traffic": pmc_traffic(traffic_kernel),
                                  "traffic_source": pmc_traffic(traffic_kernel, with_source=True)[1],
                                  "traffic_is": "HBM bytes per launch of the plain entry point (the other two: profiles/r05_notes.md, r06_notes.md)" if entries else "HBM bytes per launch",
                                  "mfma_busy_3x3": mfma_busy_3x3(),
                                  "algorithmic_bytes_per_launch": top["alg_bytes"] / top["launches"],
                                  "launches": top["launches"], "avg_launch_us": top["avg_us"],
                                  "avg_launch_us_rocprof": rp_us, "rocprof_calls": rp_calls, "rocprof_source": rp_file,
                                  "flop_per_launch_avg": top["work"] / top["launches"],
                                  "flop_per_launch_avg_executed": top["executed"] / top["launches"],
                                  "method": "second pass of the same steps, HIP events around every launch, wgrad side stream "
                                            "serialised (the timed region overlaps it with the main chain)",
                                  "peak_note": note}
            if entries:
                # the body's entry points one by one: nominal / executed rate of each, its HIP-event and rocprofv3 averages
                rp = {r["kernel"]: r for r in (rp_rows or [])}
                result["roofline"]["entries"] = [
                    {"kernel": e["kernel"], "launches": e["launches"], "avg_launch_us": e["avg_us"],
                     "avg_launch_us_rocprof": rp.get(e["kernel"], {}).get("avg_us"), "rocprof_calls": rp.get(e["kernel"], {}).get("calls"),
                     "flop_per_launch_avg": e["work"] / e["launches"],
                     "frac": round(e["work"] / (e["total_ms"] * 1e-3) / 1e12 / peak, 4),
                     "frac_algorithmic_executed": round(e["executed"] / (e["total_ms"] * 1e-3) / 1e12 / peak, 4)} for e in entries]
                result["roofline"]["entries_note"] = (
                    "one kernel body, three entry points (plain launch; ASPP's four forward branches as one grouped launch; ASPP's four "
                    "input gradients as one multi-source launch).  `frac` above = sum of their nominal FLOPs / sum of their time: the set "
                    "of convolutions that rounds 1-4 reported under the single class.  avg_launch_us_rocprof = sum of the three rows' "
                    "total time / sum of their calls in rocprof_source")
            result["kernels"] = [{k: (round(v / 1e12, 3) if k in ("work", "executed") else (round(v / 1e9, 3) if k == "alg_bytes" else v))
                                  for k, v in kk.items()} |
                                 {"tflops": round(kk["executed"] / (kk["total_ms"] * 1e-3) / 1e12, 3),
                                  "tflops_nominal": round(kk["work"] / (kk["total_ms"] * 1e-3) / 1e12, 3)} for kk in per_class]
            result["model_tflops_nominal"] = round(value * GFLOP_PER_IMG_256 * (S / 256) ** 2 / 1e3, 3)

    if world == 1 and cfg == "cfg2" and not args.no_full_width and not args.opt and not use_graph:
        # the same step at full operand width, timed here (not only in builder runs under profiles/): bf16x3 = all 24 bits
        # of both operands on the 16-bit matrix core (six MFMAs per product), fp32 = the exact-fp32 MFMA kernels everywhere
        fw = {}
        for name, opts in (("bf16x3", {"conv_arith": 0}), ("fp32_mfma", {"conv_split": 0, "wgrad_split": 0})):
            for k, v in opts.items():
                ops.set_option(k, v)
            for _ in range(2):
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                step()
            torch.cuda.synchronize()
            d = (time.perf_counter() - t0) / 5
            fw[name] = {"value": round(B / d, 1), "unit": "img/s", "ms_per_step": round(d * 1e3, 3), "steps": 5, "warmup": 2,
                        "options": opts}
            for k in opts:
                ops.set_option(k, 1)
        fw["note"] = ("same model / batch / step function as `value`, 2 warm-up + 5 timed steps each; `value` itself uses fp16x2 "
                      "(22 significant bits per operand, three MFMAs per product)")
        result["full_width"] = fw
    if world > 1:
        dist.barrier()
    if rank == 0:
        secondary = world == 1 and cfg == "cfg2"
        if secondary and not args.no_cam:
            log("cam bench")
            result["cam"] = cam_bench(device, roofline=not args.no_roofline)
            result["ncut"] = ncut_bench(device)
        if world == 1 and not args.no_cpu_baseline:
            # oracle twin of the extra loss terms of cfg3 / cfg5 (cfg4's CPU leg is the cfg2 step + the CAM legs)
            cpu_extra = None
            if cfg in ("cfg3", "cfg5"):
                import oracle
                o_ncut = oracle.LocalNormalizedCutLoss(0.1, 5)
                o_bnd = oracle.ConstrainToBoundaryLossSingle(0.1, 5, 5)
                if cfg == "cfg3":
                    cpu_extra = lambda o, i: 0.1 * o_ncut(o, i)                                            # noqa: E731
                else:
                    cpu_extra = lambda o, i: 0.1 * o_ncut(o, i) + 0.1 * torch.stack(                       # noqa: E731
                        [o_bnd(torch.softmax(o[b], 0), i[b]) for b in range(o.shape[0])]).mean()
            result["cpu_baseline"] = cpu_baseline(B, S, S, cpu_extra, steps=3 if cfg == "cfg2" else 1)
            if secondary and not args.no_cam:
                result["cam"]["cpu_baseline"] = cam_cpu_baselines()
                result["ncut"]["cpu_baseline"] = ncut_cpu_baseline(32, 256, 256)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
