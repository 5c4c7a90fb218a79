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
    for it in range(5):
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
    def shard_sum(with_aux):
        total = None
        for r in range(2):
            m = _TwoHeads()
            main, aux = m(x[r * 2:(r + 1) * 2])
            loss = main.pow(2).mean() + (aux.pow(2).mean() if with_aux else 0.0)
            loss.backward()
            flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).flatten() for p in m.parameters()])
            total = flat if total is None else total + flat
        return total

    n_aux = sum(p.numel() for p in _TwoHeads().aux.parameters())
    plain, with_aux = shard_sum(False), shard_sum(True)
    for it in (0, 2, 3, 4):
        mine = torch.cat([got["grads"][it][o:o + n] for o, n in zip(got["offsets"], got["numels"])])
        if it < 3:
            assert torch.allclose(mine, plain, rtol=1e-5, atol=1e-7), it
        else:
            # the firing set had been the same for three steps: the control exchange runs one step behind (dp.py).  The
            # head's gradient of step 3 is held back (not applied unreduced) and joins step 4's reduction - nothing is lost
            assert torch.allclose(mine[:-n_aux], with_aux[:-n_aux] if it == 3 else plain[:-n_aux], rtol=1e-5, atol=1e-7), it
            want_aux = torch.zeros(n_aux) if it == 3 else with_aux[-n_aux:]
            assert torch.allclose(mine[-n_aux:], want_aux, rtol=1e-5, atol=1e-7), it
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
    model(x[2:]).pow(2).mean().backward()                # second backward without no_sync: recorded by the hook ...
    try:
        red.wait()                                       # ... and refused by every rank together, before any Adam step
    except RuntimeError as e:
        raised = "no_sync" in str(e)
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


# ---------------------------------------------------------------------------------------------------------------
# world 8: the bucket order, the agreed bitmap with one rank firing a different set, unequal
# shards, no_sync, an error on ONE rank raised by all, and the control exchange leaving the step's critical path.
_PLAN8 = [0, 0, 0, 0, 0, 3, 0, 0, 0, 0, 5, 0]        # per step: which rank (if non-zero) ALSO uses the extra head


def _worker_world8(rank, world, port, out):
    _env(rank, world, port)
    from weaklysuperviseddl_amd.dp import init_distributed, GradBucketReducer
    from weaklysuperviseddl_amd.optim import FlatAdam
    init_distributed(backend="gloo")
    model = _Branchy()
    opt = FlatAdam(list(model.parameters()), lr=1e-3)
    red = GradBucketReducer(opt, num_buckets=3)
    # unequal shards: rank r holds 1 + (r % 3) images
    sizes = [1 + (r % 3) for r in range(world)]
    lo = sum(sizes[:rank])
    x = torch.randn(sum(sizes), 3, 8, 8, generator=torch.Generator().manual_seed(31))[lo:lo + sizes[rank]]
    grads, steady, order = [], [], []
    launch = red._launch
    red._launch = lambda b, early=False: (order.append(b), launch(b, early))[1]
    for it, who in enumerate(_PLAN8):
        opt.zero_grad()
        order.clear()
        use_extra = who != 0 and rank == who
        if it == 8:                                       # gradient accumulation: two micro-batches, the first inside no_sync
            with red.no_sync():
                model(x, use_extra).backward()
        model(x, use_extra).backward()
        red.wait()
        grads.append(opt.flat_grad.clone())
        steady.append(red._steady)
        assert order == sorted(order, reverse=True) and len(order) == len(red.bucket_size), order   # last bucket first, all of them
    stats = (red.control_exchanges_blocking, red.control_exchanges_async)
    for _ in range(4):                                    # plain steps: the firing set settles, the exchange goes asynchronous again
        opt.zero_grad()
        model(x, False).backward()
        red.wait()
    assert red._steady
    # an error on ONE rank (a second backward without no_sync on rank 2) is raised by every rank from wait()
    opt.zero_grad()
    model(x, False).backward()
    if rank == 2:
        model(x, False).backward()
    msgs, waits = [], 0
    for _ in range(3):                                    # steady mode: the flag arrives with the NEXT step's wait()
        try:
            waits += 1
            red.wait()
            opt.zero_grad()
            model(x, False).backward()
        except RuntimeError as e:
            msgs.append(str(e))
            break
    gathered = [None] * world
    dist.all_gather_object(gathered, (len(msgs), "no_sync" in (msgs[0] if msgs else ""), waits))
    if rank == 0:
        torch.save({"grads": grads, "steady": steady, "stats": stats, "raised": gathered, "sizes": sizes,
                    "offsets": opt.offsets, "numels": [p.numel() for p in opt.params]}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_world8_bucket_order_agreement_and_steady_control_exchange(tmp_path):
    world = 8
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker_world8, args=(world, _free_port(), out), nprocs=world, join=True)
    got = torch.load(out)
    sizes = got["sizes"]
    x = torch.randn(sum(sizes), 3, 8, 8, generator=torch.Generator().manual_seed(31))

    def shard_sum(it, passes):
        total = None
        for r in range(world):
            lo = sum(sizes[:r])
            m = _Branchy()
            for _ in range(passes):
                m(x[lo:lo + sizes[r]], _PLAN8[it] != 0 and r == _PLAN8[it]).backward()
            flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).flatten() for p in m.parameters()])
            total = flat if total is None else total + flat
        return total

    def mine(it):
        return torch.cat([got["grads"][it][o:o + n] for o, n in zip(got["offsets"], got["numels"])])

    first_extra = sum(got["numels"][:-2])                 # the extra head's weight + bias: the last two parameters
    for it in range(len(_PLAN8)):
        want = shard_sum(it, 2 if it == 8 else 1)
        if _PLAN8[it] != 0:
            # the deviating rank's gradient for the cut-out head is HELD BACK in the steady step ...
            assert torch.allclose(mine(it)[:first_extra], want[:first_extra], rtol=1e-5, atol=1e-7), it
            assert mine(it)[first_extra:].abs().max() == 0, it
        elif it > 0 and _PLAN8[it - 1] != 0:
            # ... and joins the next step's reduction, which every rank runs in blocking mode again
            prev = shard_sum(it - 1, 1)
            assert torch.allclose(mine(it)[:first_extra], want[:first_extra], rtol=1e-5, atol=1e-7), it
            assert torch.allclose(mine(it)[first_extra:], prev[first_extra:], rtol=1e-5, atol=1e-7), it
        else:
            assert torch.allclose(mine(it), want, rtol=1e-5, atol=1e-7), it
    # the control exchange blocks for the first STEADY_AFTER steps, then goes asynchronous; a deviation brings every
    # rank back to blocking mode one step later, until the set has settled again
    print("steady per step:", got["steady"], "exchanges (blocking, async):", got["stats"])
    # (flag recorded after each wait(): steps 0-2 agree -> asynchronous from step 3; rank 3 deviates in step 5, seen by all in
    # step 6 -> blocking again; the set of step 6 holds the extra head, steps 7-9 settle -> asynchronous in step 10, where
    # rank 5 deviates, seen in step 11)
    assert got["steady"] == [False, False, True, True, True, True, False, False, False, True, True, False], got["steady"]
    assert got["stats"] == (8, 4), got["stats"]                         # (blocking, asynchronous) exchanges over the 12 steps
    assert all(n == 1 and w == 2 for n, _, w in got["raised"]), got["raised"]     # every rank stopped, in the same (second) wait()
    assert got["raised"][2][:2] == (1, True)                            # the rank at fault names the cause
