"""world_size-2 gloo test of the data-parallel path (CPU): bucketed gradient all-reduce from
post-accumulate-grad hooks, 1/world folded into the optimiser's grad_scale, result = mean of the
per-shard gradients."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _model():
    torch.manual_seed(0)
    return nn.Sequential(nn.Conv2d(3, 8, 3, padding=1), nn.ReLU(), nn.Conv2d(8, 8, 3, padding=1), nn.ReLU(),
                         nn.Conv2d(8, 2, 1), nn.Conv2d(2, 2, 1))


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from weaklysuperviseddl_amd.dp import init_distributed, GradBucketReducer
    from weaklysuperviseddl_amd.optim import FlatAdam
    torch.set_num_threads(1)
    r, _, w = init_distributed(backend="gloo")
    assert (r, w) == (rank, world)
    model = _model()
    opt = FlatAdam(list(model.parameters()), lr=1e-3)
    red = GradBucketReducer(opt, num_buckets=3)
    assert abs(opt.grad_scale - 1.0 / world) < 1e-12 and len(red.bucket_size) >= 2
    g = torch.Generator().manual_seed(100)
    x = torch.randn(4, 3, 8, 8, generator=g)[rank * 2:(rank + 1) * 2]       # this rank's shard
    for it in range(2):                                                     # second pass checks the re-arm
        opt.zero_grad()
        model(x).pow(2).mean().backward()
        launched_early = sum(red._launched)
        red.wait()
    if rank == 0:
        torch.save({"grad": opt.flat_grad.clone(), "launched_early": launched_early, "offsets": opt.offsets,
                    "numels": [p.numel() for p in opt.params]}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_allreduce_world2(tmp_path):
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    # single-process reference: per-shard gradients summed (the reducer sums; Adam applies 1/world)
    g = torch.Generator().manual_seed(100)
    x = torch.randn(4, 3, 8, 8, generator=g)
    total = None
    for r in range(2):
        m = _model()
        m(x[r * 2:(r + 1) * 2]).pow(2).mean().backward()
        flat = torch.cat([p.grad.flatten() for p in m.parameters()])
        total = flat if total is None else total + flat
    mine = torch.cat([got["grad"][o:o + n] for o, n in zip(got["offsets"], got["numels"])])
    assert torch.allclose(mine, total, rtol=1e-5, atol=1e-7)
    assert got["launched_early"] >= 1      # at least one bucket went out from a backward hook, before wait()


class _TwoHeads(nn.Module):
    """Trunk + main head + an auxiliary head whose output no loss uses (the DeepLabV3 aux classifier's situation)."""

    def __init__(self):
        super().__init__()
        torch.manual_seed(1)
        self.trunk = nn.Sequential(nn.Conv2d(3, 8, 3, padding=1), nn.ReLU(), nn.Conv2d(8, 8, 3, padding=1), nn.ReLU())
        self.head = nn.Conv2d(8, 2, 1)
        self.aux = nn.Sequential(nn.Conv2d(8, 8, 3, padding=1), nn.Conv2d(8, 3, 1))

    def forward(self, x):
        f = self.trunk(x)
        return self.head(f), self.aux(f)


def _worker_unused(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from weaklysuperviseddl_amd.dp import init_distributed, GradBucketReducer
    from weaklysuperviseddl_amd.optim import FlatAdam
    torch.set_num_threads(1)
    init_distributed(backend="gloo")
    model = _TwoHeads()
    opt = FlatAdam(list(model.parameters()), lr=1e-3)
    red = GradBucketReducer(opt, num_buckets=2)
    g = torch.Generator().manual_seed(7)
    x = torch.randn(4, 3, 8, 8, generator=g)[rank * 2:(rank + 1) * 2]
    grads, early, excluded = [], [], []
    for it in range(4):
        opt.zero_grad()
        main, aux = model(x)
        loss = main.pow(2).mean()
        if it == 3:
            loss = loss + aux.pow(2).mean()       # the "unused" head turns up after having been cut out
        loss.backward()
        early.append(sum(red._launched))
        excluded.append(len(red._excluded))
        red.wait()
        grads.append(opt.flat_grad.clone())
    if rank == 0:
        torch.save({"grads": grads, "early": early, "excluded": excluded, "offsets": opt.offsets,
                    "numels": [p.numel() for p in opt.params]}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_unused_trailing_parameters_do_not_hold_the_last_bucket_back(tmp_path):
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker_unused, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    g = torch.Generator().manual_seed(7)
    x = torch.randn(4, 3, 8, 8, generator=g)
    for it in (0, 2, 3):
        total = None
        for r in range(2):
            m = _TwoHeads()
            main, aux = m(x[r * 2:(r + 1) * 2])
            loss = main.pow(2).mean() + (aux.pow(2).mean() if it == 3 else 0.0)
            loss.backward()
            flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).flatten() for p in m.parameters()])
            total = flat if total is None else total + flat
        mine = torch.cat([got["grads"][it][o:o + n] for o, n in zip(got["offsets"], got["numels"])])
        assert torch.allclose(mine, total, rtol=1e-5, atol=1e-7), it
    # step 0 has to wait for the aux head (its bucket leaves at the join); from step 1 on the head is cut out and
    # every bucket leaves from a backward hook
    assert got["excluded"][0] == 0 and got["excluded"][1] == 4
    assert got["early"][1] > got["early"][0] and got["early"][1] == 2


# ---------------------------------------------------------------------------------------------------------------
# What keeps replicas identical when ranks differ: initial-state broadcast, an agreed firing
# bitmap (no rank-local decision about which collectives to issue), once-per-step counting, no_sync accumulation.
def _spawn(fn, tmp_path, *extra):
    out = str(tmp_path / "r0.pt")
    mp.spawn(fn, args=(2, _free_port(), out) + extra, nprocs=2, join=True)
    return torch.load(out)


def _env(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)


def _worker_state_sync(rank, world, port, out):
    _env(rank, world, port)
    from weaklysuperviseddl_amd.dp import init_distributed, GradBucketReducer, average_bn_buffers
    from weaklysuperviseddl_amd.optim import FlatAdam
    from weaklysuperviseddl_amd import ops
    init_distributed(backend="gloo")
    torch.manual_seed(1234 + rank)                       # replicas start DIFFERENT (a checkpoint loaded on one rank)
    model = nn.Sequential(nn.Conv2d(3, 4, 3, padding=1), nn.BatchNorm2d(4), nn.ReLU(), nn.Conv2d(4, 2, 1))
    model[1].running_mean.fill_(float(rank + 1))
    opt = FlatAdam(list(model.parameters()), lr=1e-3)
    GradBucketReducer(opt, num_buckets=2, modules=[model])
    flat = opt.flat_param.clone()
    rm_after_sync = model[1].running_mean.clone()
    model[1].running_mean.fill_(float(rank + 1))         # per-replica statistics drift apart during training ...
    average_bn_buffers([model])                          # ... and are averaged for the checkpoint
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    if rank == 0:
        torch.save({"flats": gathered, "rm_sync": rm_after_sync, "rm_avg": model[1].running_mean.clone(),
                    "seed_off": ops.DROPOUT_SEED_OFFSET[0]}, out)
    assert (ops.DROPOUT_SEED_OFFSET[0] != 0) == (rank != 0)
    dist.barrier()
    dist.destroy_process_group()


def test_initial_state_is_broadcast_and_bn_buffers_can_be_averaged(tmp_path):
    got = _spawn(_worker_state_sync, tmp_path)
    assert torch.equal(got["flats"][0], got["flats"][1])
    assert torch.all(got["rm_sync"] == 1.0)              # rank 0's buffers everywhere
    assert torch.all(got["rm_avg"] == 1.5)               # (1 + 2) / 2


class _Branchy(nn.Module):
    """A head that only some ranks use on some steps (data-dependent branch), and a weight used twice."""

    def __init__(self):
        super().__init__()
        torch.manual_seed(2)
        self.a = nn.Conv2d(3, 6, 3, padding=1)
        self.shared = nn.Conv2d(6, 6, 3, padding=1)
        self.head = nn.Conv2d(6, 2, 1)
        self.extra = nn.Conv2d(6, 2, 1)

    def forward(self, x, use_extra):
        f = torch.relu(self.shared(torch.relu(self.shared(torch.relu(self.a(x))))))      # `shared` fires once, used twice
        y = self.head(f).pow(2).mean()
        if use_extra:
            y = y + self.extra(f).pow(2).mean()
        return y


_PLAN = [(False, False), (False, False), (True, False), (False, True), (True, True), (False, False)]   # (rank0, rank1) per step


def _worker_branchy(rank, world, port, out):
    _env(rank, world, port)
    from weaklysuperviseddl_amd.dp import init_distributed, GradBucketReducer
    from weaklysuperviseddl_amd.optim import FlatAdam
    init_distributed(backend="gloo")
    model = _Branchy()
    opt = FlatAdam(list(model.parameters()), lr=1e-3)
    red = GradBucketReducer(opt, num_buckets=3)
    x = torch.randn(4, 3, 8, 8, generator=torch.Generator().manual_seed(11))[rank * 2:(rank + 1) * 2]
    grads = []
    for use in _PLAN:
        opt.zero_grad()
        model(x, use[rank]).backward()
        red.wait()                      # would hang (mismatched collectives) if ranks decided locally
        grads.append(opt.flat_grad.clone())
    if rank == 0:
        torch.save({"grads": grads, "offsets": opt.offsets, "numels": [p.numel() for p in opt.params]}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_ranks_that_fire_different_parameters_issue_the_same_collectives(tmp_path):
    got = _spawn(_worker_branchy, tmp_path)
    x = torch.randn(4, 3, 8, 8, generator=torch.Generator().manual_seed(11))
    for it, use in enumerate(_PLAN):
        total = None
        for r in range(2):
            m = _Branchy()
            m(x[r * 2:(r + 1) * 2], use[r]).backward()
            flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).flatten() for p in m.parameters()])
            total = flat if total is None else total + flat
        mine = torch.cat([got["grads"][it][o:o + n] for o, n in zip(got["offsets"], got["numels"])])
        assert torch.allclose(mine, total, rtol=1e-5, atol=1e-7), it


def _worker_accumulate(rank, world, port, out):
    _env(rank, world, port)
    from weaklysuperviseddl_amd.dp import init_distributed, GradBucketReducer
    from weaklysuperviseddl_amd.optim import FlatAdam
    init_distributed(backend="gloo")
    model = _model()
    opt = FlatAdam(list(model.parameters()), lr=1e-3)
    red = GradBucketReducer(opt, num_buckets=3)
    x = torch.randn(8, 3, 8, 8, generator=torch.Generator().manual_seed(21))[rank * 4:(rank + 1) * 4]
    raised = False
    for it in range(2):
        opt.zero_grad()
        with red.no_sync():
            model(x[:2]).pow(2).mean().backward()        # first micro-batch: accumulate only
        model(x[2:]).pow(2).mean().backward()            # last micro-batch: buckets leave
        red.wait()
    acc = opt.flat_grad.clone()
    opt.zero_grad()
    model(x[:2]).pow(2).mean().backward()
    try:
        model(x[2:]).pow(2).mean().backward()            # second backward without no_sync: refused, not mis-reduced
    except RuntimeError:
        raised = True
    gathered = [None] * world
    dist.all_gather_object(gathered, raised)
    if rank == 0:
        torch.save({"grad": acc, "raised": gathered, "offsets": opt.offsets, "numels": [p.numel() for p in opt.params]}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_accumulation_needs_no_sync(tmp_path):
    got = _spawn(_worker_accumulate, tmp_path)
    x = torch.randn(8, 3, 8, 8, generator=torch.Generator().manual_seed(21))
    total = None
    for r in range(2):
        m = _model()
        xs = x[r * 4:(r + 1) * 4]
        m(xs[:2]).pow(2).mean().backward()
        m(xs[2:]).pow(2).mean().backward()
        flat = torch.cat([p.grad.flatten() for p in m.parameters()])
        total = flat if total is None else total + flat
    mine = torch.cat([got["grad"][o:o + n] for o, n in zip(got["offsets"], got["numels"])])
    assert torch.allclose(mine, total, rtol=1e-5, atol=1e-7)
    assert got["raised"] == [True, True]
