#!/usr/bin/env python3
"""Known-byte-count launches for calibrating FETCH_SIZE / WRITE_SIZE on gfx950 in OUR access patterns
(MI355X_MICROARCH.md: FETCH_SIZE reads 1/2 of a 16 B/lane stream; other widths are uncalibrated).
Run under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE`; tools/pmc_summarise.py reads the CSVs.

  copy_planes_kernel : 4 B/lane loads and stores, 1 GiB in / 1 GiB out  (the conv kernels' activation access width)
  adam_kernel        : 16 B/lane, 4 x 256 MiB in / 3 x 256 MiB out
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C  # noqa: E402
import torch  # noqa: E402
from weaklysuperviseddl_amd import ops  # noqa: E402
from weaklysuperviseddl_amd._lib import lib, check  # noqa: E402

dev = torch.device("cuda:0")
n = 1 << 28                                    # 256 Mi floats = 1 GiB
src = torch.randn(n, device=dev)
dst = torch.empty_like(src)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(3):
    check(lib().wsdl_copy_planes(C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), 1, 1024, n // 1024, 0, 0, st))
m = 1 << 26                                    # 64 Mi floats = 256 MiB per buffer
p, g, a, b = (torch.randn(m, device=dev) for _ in range(4))
b.abs_()
for i in range(3):
    ops.adam_step_flat(p, g, a, b, 1e-4, 0.9, 0.999, 1e-8, i + 1)
torch.cuda.synchronize()
print("calibration launches done: copy_planes 1 GiB r / 1 GiB w; adam 1 GiB r / 0.75 GiB w")
