"""Data parallelism THROUGH THE HIP PATH: two ranks sharing one GPU over gloo
(``WSDL_DIST_BACKEND=gloo``; on an 8-GPU node the same code runs over RCCL).  Checked:
  * gradients written directly into the flat buffer by the HIP backward kernels (``grad_ready_hooks``), bucket
    all-reduces enqueued from the side stream: reduced flat_grad == sum of the per-shard single-process gradients;
  * buckets leave from backward hooks (before ``wait()``) from the second step on; replicas stay identical;
  * stage 1 (CAM -> pseudo masks) sharded round-robin over the ranks: union == single-process masks, bit-exact;
  * the alternating loop under DP with unequal shard sizes: same number of optimiser steps on every rank.
Children are fresh processes (``mp.spawn``) that pick their device before any other GPU call."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _env(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), WSDL_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.set_num_threads(2)
    torch.cuda.set_device(0)


def _seg_model(device, seed=0):
    from weaklysuperviseddl_amd import nn as wnn
    from weaklysuperviseddl_amd.TraditionalModel import build_segmentation_model
    torch.manual_seed(seed)
    model = build_segmentation_model()
    for m in model.modules():
        if isinstance(m, wnn.Dropout):
            m.p = 0.0
    return model.to(device).train()


def _shard(rank, B=2, S=64):
    import bench
    img, masks = bench.synthetic_batch(2 * B, S, S, "cpu", 7)
    return img[rank * B:(rank + 1) * B], masks[rank * B:(rank + 1) * B]


def _grads_only(model, opt, img, masks):
    from weaklysuperviseddl_amd import ops
    out = model(img)["out"]
    loss = ops.cross_entropy(out, masks)
    opt.zero_grad()
    loss.backward()
    return loss


def _worker_train(rank, world, port, out):
    _env(rank, world, port)
    import torch.distributed as dist
    from weaklysuperviseddl_amd import ops
    from weaklysuperviseddl_amd.dp import init_distributed, GradBucketReducer
    from weaklysuperviseddl_amd.TraditionalModel import train_step
    from weaklysuperviseddl_amd.TraditionalModel.SegmentationModel import make_optimizer
    dev = torch.device("cuda:0")
    init_distributed()
    assert dist.get_backend() == "gloo"
    model = _seg_model(dev, seed=rank)                 # different seeds: the reducer must broadcast rank 0's weights
    opt = make_optimizer(model, lr=1e-4)
    red = GradBucketReducer(opt, modules=[model])
    img, masks = (t.to(dev) for t in _shard(rank))
    _grads_only(model, opt, img, masks)
    early0 = sum(red._launched)
    red.wait()
    ops.join_side_stream(dev)
    torch.cuda.synchronize()
    reduced = opt.flat_grad.detach().cpu().clone()
    excluded = len(red._excluded)
    early = []
    for _ in range(3):                                 # full steps: hooks, side-stream collectives, Adam, weight prefetch
        train_step(model, opt, img, masks)
        early.append(red.last_early_launches)
    torch.cuda.synchronize()
    params = opt.flat_param.detach().cpu()
    gathered = [torch.zeros_like(params) for _ in range(world)]
    dist.all_gather(gathered, params)
    if rank == 0:
        torch.save({"reduced": reduced, "early0": early0, "early": early, "excluded": excluded, "params": gathered,
                    "buckets": len(red.bucket_size)}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_hip_dp_reduced_gradients_equal_the_sum_of_the_shard_gradients(dev, tmp_path):
    from weaklysuperviseddl_amd import ops
    from weaklysuperviseddl_amd.TraditionalModel.SegmentationModel import make_optimizer
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker_train, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    total = None
    for r in range(2):
        model = _seg_model(dev, seed=0)
        opt = make_optimizer(model, lr=1e-4)
        img, masks = (t.to(dev) for t in _shard(r))
        _grads_only(model, opt, img, masks)
        ops.join_side_stream(dev)
        torch.cuda.synchronize()
        g = opt.flat_grad.detach().cpu().clone()
        total = g if total is None else total + g
    scale = total.abs().max().item()
    assert scale > 0
    # the kernels are deterministic and the sum of two terms is order-free: equal to fp32 rounding
    assert (got["reduced"] - total).abs().max().item() <= 1e-6 * scale
    assert got["buckets"] >= 4 and got["excluded"] > 0            # the aux head was learnt to be unused
    # step 0 has to wait for the aux head (nothing is known yet); afterwards every bucket leaves from a hook
    assert got["early"][0] >= 1 and got["early"][-1] == got["buckets"]
    assert torch.equal(got["params"][0], got["params"][1])       # replicas stay bit-identical


def _worker_rccl_single(rank, world, port, out, early):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                      WSDL_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0", WSDL_EARLY_STEP=early)
    os.environ.pop("WSDL_DIST_BACKEND", None)
    torch.set_num_threads(2)
    import torch.distributed as dist
    from weaklysuperviseddl_amd.dp import init_distributed, GradBucketReducer, average_bn_buffers
    from weaklysuperviseddl_amd.TraditionalModel import train_step
    from weaklysuperviseddl_amd.TraditionalModel.SegmentationModel import make_optimizer
    dev = torch.device("cuda:0")
    init_distributed()
    assert dist.is_initialized() and dist.get_backend() == "nccl" and dist.get_world_size() == 1
    model = _seg_model(dev, seed=0)
    opt = make_optimizer(model, lr=1e-4)
    red = GradBucketReducer(opt, modules=[model])      # RCCL broadcasts of parameters / moments / buffers
    assert red.active
    img, masks = (t.to(dev) for t in _shard(0))
    losses, early = [], []
    for _ in range(3):                                 # RCCL all-reduces enqueued from the side stream by the hooks
        losses.append(float(train_step(model, opt, img, masks)))
        early.append(red.last_early_launches)
    average_bn_buffers([model])
    torch.cuda.synchronize()
    # a data-parallel process must not keep more streams busy than it has hardware queues (4): main + side + prep + RCCL's own
    # are four already, so the LayerCAM lanes beyond the second take turns on the side / prep stream instead of a fifth stream
    from weaklysuperviseddl_amd import ops
    census_before = ops.stream_census(dev)
    lanes = [ops.lane_stream(dev, i) for i in range(4)]
    census = ops.stream_census(dev)
    params = opt.flat_param.detach().cpu()             # after the three steps the plain process is compared with
    opt.time_tail = True
    train_step(model, opt, img, masks)
    opt.time_tail = False
    tail = opt.tail_ms()
    torch.save({"params": params, "losses": losses, "early": early,
                "buckets": len(red.bucket_size), "census_before": census_before, "census": census,
                "lane_ids": [st.cuda_stream for st in lanes], "tail_ms": tail}, out)
    dist.destroy_process_group()


@pytest.mark.parametrize("early", ["0", "1"])
def test_single_rank_rccl_path_equals_the_plain_step(dev, tmp_path, early):
    """(early = "1": each bucket's Adam launch and weight re-layout follow its collective on the side stream.)
    The RCCL calls themselves on the one GPU there is (WSDL_FORCE_DIST=1: backend "nccl", world 1): broadcasts,
    bucketed all-reduces launched from backward hooks on the side stream, the gloo control exchange.  A one-rank sum is
    the identity and 1/world = 1, so three steps must reproduce the plain single-process steps bit for bit."""
    from weaklysuperviseddl_amd.TraditionalModel import train_step
    from weaklysuperviseddl_amd.TraditionalModel.SegmentationModel import make_optimizer
    out = str(tmp_path / "single.pt")
    mp.spawn(_worker_rccl_single, args=(1, _free_port(), out, early), nprocs=1, join=True)
    got = torch.load(out)
    model = _seg_model(dev, seed=0)
    opt = make_optimizer(model, lr=1e-4)
    img, masks = (t.to(dev) for t in _shard(0))
    losses = [float(train_step(model, opt, img, masks)) for _ in range(3)]
    torch.cuda.synchronize()
    assert got["losses"] == losses
    assert torch.equal(got["params"], opt.flat_param.detach().cpu())
    assert got["buckets"] >= 5 and got["early"][-1] == got["buckets"]   # every bucket left from a hook once the aux head was known
    # streams of the data-parallel process: never more than the hardware queues; lanes 2, 3 reuse the side / prep stream
    assert got["census_before"]["rccl"] == 1 and got["census_before"]["total"] <= 4, got["census_before"]
    assert got["census"]["total"] <= got["census"]["hw_queues"] == 4, got["census"]
    assert got["census"]["lanes"] == 0 and len(set(got["lane_ids"])) == 2, (got["census"], got["lane_ids"])
    assert got["tail_ms"] is not None and 0.0 < got["tail_ms"] < 50.0
    # a plain process has room for one lane stream of its own (main + side + prep + one)
    from weaklysuperviseddl_amd import ops
    st = [ops.lane_stream(dev, i).cuda_stream for i in range(5)]
    c = ops.stream_census(dev)
    assert c["rccl"] == 0 and c["total"] <= 4 and c["lanes"] == 1 and len(set(st)) == 3, (c, st)


def test_early_segment_steps_equal_the_single_launch_step(dev):
    """FlatAdam stepped in segments as backward completes them (each followed by the re-layout of its convolution weights
    into the spare buffers) == one Adam launch after backward, bit for bit, over four steps; the segments are small at
    the front of the buffer and every one but the first leaves before step() from the second step on."""
    from weaklysuperviseddl_amd.TraditionalModel import train_step
    from weaklysuperviseddl_amd.TraditionalModel.SegmentationModel import make_optimizer
    img, masks = (t.to(dev) for t in _shard(0))
    out = {}
    for early in (False, True):
        model = _seg_model(dev, seed=3)
        opt = make_optimizer(model, lr=1e-4, early_step=early)
        losses, left_early = [], []
        for _ in range(4):
            stepped_before = []
            opt.pre_step_hook = lambda: stepped_before.append(sum(opt._stepped))
            losses.append(float(train_step(model, opt, img, masks)))
            left_early.append(stepped_before[0])
        torch.cuda.synchronize()
        out[early] = (losses, opt.flat_param.detach().cpu().clone(), opt.exp_avg_sq.detach().cpu().clone(), left_early)
        sizes = [hi - lo for lo, hi in opt.segments]
    assert out[True][0] == out[False][0]
    assert torch.equal(out[True][1], out[False][1]) and torch.equal(out[True][2], out[False][2])
    assert len(sizes) >= 5 and sizes[0] < sizes[1] < sizes[2] and sizes[0] <= 600_000
    assert out[False][3] == [0, 0, 0, 0]
    assert out[True][3][0] == 0 and out[True][3][-1] >= len(sizes) - 1


def _cam_loader(n_batches=4, B=2):
    g = torch.Generator().manual_seed(41)
    return [(torch.rand(B, 3, 224, 224, generator=g), ((torch.arange(B) + 5 * j) % 37, None)) for j in range(n_batches)]


def _cam_generator(device):
    from weaklysuperviseddl_amd.TraditionalModel import FrozenResNetCAM, LayerCAMGenerator
    torch.manual_seed(0)
    model = FrozenResNetCAM(37)
    g = torch.Generator().manual_seed(3)
    for m in model.modules():
        if hasattr(m, "running_mean"):
            m.running_mean.copy_(torch.randn(m.num_features, generator=g) * 0.1)
            m.running_var.copy_(torch.rand(m.num_features, generator=g) + 0.5)
    return LayerCAMGenerator(model.to(device).eval(), ["layer3", "layer4"])


def _worker_stage1(rank, world, port, out):
    _env(rank, world, port)
    import torch.distributed as dist
    from weaklysuperviseddl_amd.dp import init_distributed
    from weaklysuperviseddl_amd.TraditionalModel import generate_pseudo_masks
    init_distributed()
    gen = _cam_generator(torch.device("cuda:0"))
    # device_batch=0: one launch sequence per loader batch, so a rank's results do not depend on which other batches it holds
    # (merged device batches - the default - carry their own amax scales: equal to fp32 noise only)
    generate_pseudo_masks(_cam_loader(), gen, cam_thresh=0.3, write_png=False, max_images=7, rank=rank, world=world, device_batch=0)
    mine = (generate_pseudo_masks.last_ids, generate_pseudo_masks.last_masks)
    both = [None] * world
    dist.all_gather_object(both, mine)
    if rank == 0:
        torch.save(both, out)
    dist.barrier()
    dist.destroy_process_group()


def test_stage1_sharded_over_ranks_equals_single_process(dev, tmp_path):
    from weaklysuperviseddl_amd.TraditionalModel import generate_pseudo_masks
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker_stage1, args=(2, _free_port(), out), nprocs=2, join=True)
    both = torch.load(out, weights_only=False)
    generate_pseudo_masks(_cam_loader(), _cam_generator(dev), cam_thresh=0.3, write_png=False, max_images=7, device_batch=0)
    ids, masks = generate_pseudo_masks.last_ids, generate_pseudo_masks.last_masks
    assert ids == list(range(7))                                   # the cap cuts the last batch (PsuedoMasks.py:49)
    assert sorted(both[0][0] + both[1][0]) == ids and not set(both[0][0]) & set(both[1][0])
    assert both[0][0] == [0, 1, 4, 5] and both[1][0] == [2, 3, 6]  # batch j -> rank j % 2
    union = {i: m for r in range(2) for i, m in zip(*both[r])}
    for i, m in zip(ids, masks):
        assert np.array_equal(union[i], m), i                      # bit-exact: same batches, deterministic kernels
    assert sum(int(m.sum()) for m in masks) > 0


def test_stage_handoff_equals_the_png_round_trip(dev, tmp_path):
    """In-memory stage-1 -> stage-2 hand-off vs the reference's files (PsuedoMasks.py:68-74 -> SegmentationDataset.py)."""
    from weaklysuperviseddl_amd.TraditionalModel import (generate_pseudo_masks, stage_handoff, PseudoSegmentationDataset)
    gen = _cam_generator(dev)
    loader = _cam_loader(2, 3)
    img_dir, mask_dir = generate_pseudo_masks(loader, gen, cam_thresh=0.3, out_root=str(tmp_path), run_id="t",
                                              write_png=True, keep_images=True)
    masks, images = generate_pseudo_masks.last_masks, torch.stack(generate_pseudo_masks.last_images)
    im256, m256 = stage_handoff(images, masks, (256, 256), dev)
    ds = PseudoSegmentationDataset(img_dir, mask_dir, transform=True, return_name=True)
    order = sorted(range(6), key=lambda i: f"{i}.png")             # sorted(listdir): "0.png", "1.png", ...
    for k, i in enumerate(order):
        img, mask, name = ds[k]
        assert name == f"{i}.png"
        assert torch.equal(m256[i].cpu().long(), mask)             # NEAREST 224 -> 256 of the {0,255} PNG
        # PIL resamples 8-bit values in fixed point: one 8-bit level (1/255/std <= 0.0175) at most
        assert (im256[i].cpu() - img).abs().max().item() <= 1.0 / 255 / 0.224 + 1e-5
        assert ((im256[i].cpu() - img).abs() > 1e-5).float().mean().item() < 0.02
    assert set(np.unique(m256.cpu().numpy())) <= {0, 255}


def _toy_dataset(device, n, seed):
    from conftest import smooth_image
    from weaklysuperviseddl_amd.TraditionalModel import InMemoryPseudoDataset
    img = smooth_image(n, 64, 64, seed)
    masks = (img[:, 0] > 0.5).to(torch.uint8) * 255
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    return InMemoryPseudoDataset(((img - mean) / std).to(device), masks.to(device))


def test_alternating_loop_chains_refinement_in_memory(dev):
    """Outer loop of reference AlternatingDirectionCutLoss.py:791-818 on a toy shard: train <-> chained refinement."""
    from weaklysuperviseddl_amd.TraditionalModel import (run_alternating_training, refine_dataset,
                                                         refine_pseudo_masks_batched)
    from weaklysuperviseddl_amd.TraditionalModel.SegmentationModel import make_optimizer
    model = _seg_model(dev)
    opt = make_optimizer(model, lr=1e-4)
    ds = _toy_dataset(dev, 6, 3)
    m0 = ds.masks.clone()
    logs = []
    hist = run_alternating_training(model, opt, ds, num_alternations=2, epochs_per_round=2, refine_repeats=2,
                                    first_batch_size=4, later_batch_size=6, refine_chunk=4, log=logs.append,
                                    refine_kwargs=dict(lr=0.5, num_steps=5))
    assert len(hist) == 2 and all(len(h["losses"]) == 2 for h in hist)
    assert all(np.isfinite(t.item()) for h in hist for t in h["losses"])
    assert opt.step_count == 2 * 2 + 2 * 1                          # 6 images: batches of 4+2, then one batch of 6
    assert set(np.unique(ds.masks.cpu().numpy())) <= {0, 255} and not torch.equal(ds.masks, m0)
    assert logs[-1] == "Alternating training and pseudo mask updates completed."
    # refine_dataset == the reference's chain: pass r+1 starts from pass r's thresholded {0,255} masks
    ds2 = _toy_dataset(dev, 6, 3)
    want = ds2.masks.clone()
    for _ in range(2):
        r = refine_pseudo_masks_batched(model, ds2.images, want, threshold=0.3, lr=0.5, num_steps=5, lambda_boundary=0.1)
        want = (r > 0).to(torch.uint8) * 255
    refine_dataset(model, ds2, repeats=2, chunk=4, threshold=0.3, lr=0.5, num_steps=5, lambda_boundary=0.1)
    assert torch.equal(ds2.masks, want)


def _worker_alternating(rank, world, port, out):
    _env(rank, world, port)
    import torch.distributed as dist
    from weaklysuperviseddl_amd.dp import init_distributed, GradBucketReducer
    from weaklysuperviseddl_amd.TraditionalModel import run_alternating_training
    from weaklysuperviseddl_amd.TraditionalModel.SegmentationModel import make_optimizer
    dev = torch.device("cuda:0")
    init_distributed()
    model = _seg_model(dev)
    opt = make_optimizer(model, lr=1e-4)
    GradBucketReducer(opt, modules=[model])
    ds = _toy_dataset(dev, 5 if rank == 0 else 3, 20 + rank)        # unequal shards: 3 vs 2 batches of 2
    run_alternating_training(model, opt, ds, num_alternations=1, epochs_per_round=2, refine_repeats=1,
                             first_batch_size=2, refine_chunk=4, log=None, refine_kwargs=dict(num_steps=2))
    torch.cuda.synchronize()
    params = opt.flat_param.detach().cpu()
    gathered = [torch.zeros_like(params) for _ in range(world)]
    dist.all_gather(gathered, params)
    steps = [None] * world
    dist.all_gather_object(steps, opt.step_count)
    if rank == 0:
        torch.save({"params": gathered, "steps": steps}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_alternating_loop_under_dp_takes_the_same_steps_on_every_rank(dev, tmp_path):
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker_alternating, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out, weights_only=False)
    assert got["steps"] == [2, 2]                                  # min(2, 1) batches per epoch x 2 epochs
    assert torch.equal(got["params"][0], got["params"][1])


def _worker_plan_dp(rank, world, port, out, backend, planned):
    """Ten data-parallel steps: with ``planned`` the steps after the reducer has settled are launch-plan replays whose
    collectives / wait() run as host sections."""
    if isinstance(planned, (tuple, list)):
        planned = planned[rank]             # per rank: the ranks of a job need not agree on replaying
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      HSA_ENABLE_IPC_MODE_LEGACY="0", WSDL_PLAN_STEP="1" if planned else "0")
    if backend == "gloo":
        os.environ["WSDL_DIST_BACKEND"] = "gloo"
    else:
        os.environ.pop("WSDL_DIST_BACKEND", None)
        os.environ["WSDL_FORCE_DIST"] = "1"
    torch.set_num_threads(2)
    torch.cuda.set_device(0)
    import torch.distributed as dist
    from weaklysuperviseddl_amd.dp import init_distributed, GradBucketReducer
    from weaklysuperviseddl_amd.TraditionalModel import train_step
    from weaklysuperviseddl_amd.TraditionalModel.SegmentationModel import make_optimizer
    dev = torch.device("cuda:0")
    init_distributed()
    model = _seg_model(dev, seed=rank)
    opt = make_optimizer(model, lr=1e-4)
    red = GradBucketReducer(opt, modules=[model])
    batches = [tuple(t.to(dev) for t in _shard(rank if world > 1 else 0)), tuple(t.to(dev).flip(0).contiguous() for t in _shard(rank if world > 1 else 0))]
    losses = []
    for i in range(10):
        losses.append(float(train_step(model, opt, *batches[i % 2])))
    torch.cuda.synchronize()
    st = next(iter(opt.__dict__.get("_wsdl_planned", {}).values()), None)
    params = opt.flat_param.detach().cpu()
    gathered = [torch.zeros_like(params) for _ in range(world)]
    if world > 1:
        dist.all_gather(gathered, params)
    else:
        gathered = [params]
    if rank == 0:
        torch.save({"params": gathered, "losses": losses, "replays": 0 if st is None else st.replays,
                    "disabled": None if st is None else st.disabled,
                    "sections": 0 if st is None or st.plan is None else len(st.plan.sections),
                    "buckets": len(red.bucket_size), "early": red.last_early_launches,
                    "ctl": (red.control_exchanges_blocking, red.control_exchanges_async)}, out)
    if world > 1:
        dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("backend,world", [("nccl", 1), ("gloo", 2)])
def test_data_parallel_steps_replayed_from_a_plan_equal_the_eager_ones(dev, tmp_path, backend, world):
    """Under a GradBucketReducer the bucket all-reduces (launched from backward hooks in an eager step) and ``wait()`` become
    HOST SECTIONS of the step's launch plan: a replay issues the same collectives at the same places of the launch sequence.
    One rank over RCCL (the collectives themselves on the one GPU there is) and two ranks over gloo sharing the GPU: ten
    steps with replays equal ten eager data-parallel steps bit for bit, on every replica."""
    res = {}
    for planned in (False, True):
        out = str(tmp_path / f"plan_dp_{int(planned)}.pt")
        mp.spawn(_worker_plan_dp, args=(world, _free_port(), out, backend, planned), nprocs=world, join=True)
        res[planned] = torch.load(out)
    a, b = res[False], res[True]
    assert b["disabled"] is None and b["replays"] >= 3, (b["disabled"], b["replays"])
    assert b["sections"] == b["buckets"] + 1 or b["sections"] >= 2, b["sections"]           # every non-empty bucket + wait()
    assert a["losses"] == b["losses"]
    for pa, pb in zip(a["params"], b["params"]):
        assert torch.equal(pa, pb)
    assert all(torch.equal(b["params"][0], p) for p in b["params"])                         # replicas identical
    assert b["early"] == a["early"]


def test_ranks_that_disagree_on_replaying_stay_in_step(dev, tmp_path):
    """Rank 0 records, verifies and replays a plan while rank 1 stays eager (its plans are switched off): recording's
    verification steps are rank-local (no collective, no control exchange), and an eager and a replayed step issue the same
    collectives in the same order - the job neither hangs nor diverges, and equals the all-eager job bit for bit."""
    res = {}
    for name, planned in (("eager", (False, False)), ("mixed", (True, False))):
        out = str(tmp_path / f"mixed_{name}.pt")
        mp.spawn(_worker_plan_dp, args=(2, _free_port(), out, "gloo", planned), nprocs=2, join=True)
        res[name] = torch.load(out)
    a, b = res["eager"], res["mixed"]
    assert b["disabled"] is None and b["replays"] >= 3, (b["disabled"], b["replays"])
    assert a["losses"] == b["losses"]
    for pa, pb in zip(a["params"], b["params"]):
        assert torch.equal(pa, pb)
    assert torch.equal(b["params"][0], b["params"][1])


def test_deferred_slab_reductions_under_the_bucket_reducer(dev, tmp_path):
    """WSDL_WGRAD_DEFER=1 under data parallelism (one rank over RCCL): a bucket's pending slab reductions are flushed - one launch -
    right before its all-reduce is enqueued (dp.GradBucketReducer._launch), eagerly and as part of a replayed plan.  Ten
    steps equal the default's (per-layer reductions) bit for bit."""
    res = {}
    old = os.environ.get("WSDL_WGRAD_DEFER")
    try:
        for name, env in (("default", "0"), ("deferred", "1")):
            os.environ["WSDL_WGRAD_DEFER"] = env
            out = str(tmp_path / f"defer_dp_{name}.pt")
            mp.spawn(_worker_plan_dp, args=(1, _free_port(), out, "nccl", True), nprocs=1, join=True)
            res[name] = torch.load(out)
    finally:
        if old is None:
            os.environ.pop("WSDL_WGRAD_DEFER", None)
        else:
            os.environ["WSDL_WGRAD_DEFER"] = old
    a, b = res["default"], res["deferred"]
    assert b["disabled"] is None and b["replays"] >= 3, (b["disabled"], b["replays"])
    assert a["losses"] == b["losses"]
    assert torch.equal(a["params"][0], b["params"][0])
