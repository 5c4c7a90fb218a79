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
