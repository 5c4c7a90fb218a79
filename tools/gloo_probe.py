"""Bare gloo all-reduce of device tensors, two ranks on one GPU: what the rehearsal's collectives cost by themselves.
    python tools/gloo_probe.py"""
import os
import sys
import time

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def worker(rank, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE="2")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=2)
    for mb in (1, 5, 48, 158):
        t = torch.ones(mb * 250_000, device="cuda")
        for _ in range(2):
            dist.all_reduce(t)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        c0 = time.process_time()
        for _ in range(3):
            w = dist.all_reduce(t, async_op=True)
            w.wait()
        torch.cuda.synchronize()
        if rank == 0:
            print(f"{mb:4d} MB: {(time.perf_counter() - t0) / 3 * 1e3:8.1f} ms wall, {(time.process_time() - c0) / 3 * 1e3:8.1f} ms cpu",
                  file=sys.stderr, flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    mp.spawn(worker, args=(29511,), nprocs=2, join=True)
