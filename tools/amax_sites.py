#!/usr/bin/env python3
"""Which tensors of a training step still need a separate amax read pass (no producing kernel published their maximum)?
Prints the call sites of ops.amax_of that launch wsdl_amax during one step, with tensor shapes."""
import collections
import os
import sys
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from weaklysuperviseddl_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
model, opt, step, _ = bench.build_workload(sys.argv[1] if len(sys.argv) > 1 else "cfg2", 16, 256, dev, 0)
for _ in range(3):
    step()
sites = collections.Counter()
real = ops.amax_of


def spy(t, needed=True):
    a = getattr(t, "_wsdl_amax", None)
    stale = a is not None and getattr(t, "_wsdl_amax_version", t._version) != t._version
    if needed and (a is None or stale):
        fr = [f for f in traceback.extract_stack()[:-1] if "weaklysuperviseddl_amd" in f.filename][-3:]
        sites[(tuple(t.shape), "stale" if stale else "none", " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in reversed(fr)))] += 1
    return real(t, needed)


ops.amax_of = spy
step()
torch.cuda.synchronize()
for (shape, why, where), n in sorted(sites.items(), key=lambda kv: -kv[1]):
    print(f"{n:3d} x {str(shape):24s} {why:5s} {where}")
print("total", sum(sites.values()))
