"""The yardstick of __graft_entry__.smoke() over seeds: HIP / fp32-oracle ratio of the median-over-parameters relative L2
distance to float64 gradients, 2 x 64 x 64 train-mode step (10 seeds: 0.70-1.26, median 1.04).   python tools/smoke_ratio.py"""
import sys, copy
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
import oracle
from weaklysuperviseddl_amd import nn as wnn, ops
from weaklysuperviseddl_amd.TraditionalModel import build_segmentation_model
dev = torch.device('cuda:0')
def rel_l2(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return ((a - b).norm() / (b.norm() + 1e-300)).item()
rs = []
for seed in range(10):
    torch.manual_seed(seed)
    ref = oracle.build_segmentation_model()
    mine = build_segmentation_model(); mine.load_state_dict(ref.state_dict())
    for m in ref.modules():
        if isinstance(m, torch.nn.Dropout): m.p = 0.0
    for m in mine.modules():
        if isinstance(m, wnn.Dropout): m.p = 0.0
    mine = mine.to(dev).train(); ref.train()
    x = torch.randn(2, 3, 64, 64); masks = (torch.rand(2, 64, 64) > 0.5).long()
    ref64 = copy.deepcopy(ref).double()
    F.cross_entropy(ref(x)['out'], masks).backward()
    F.cross_entropy(ref64(x.double())['out'], masks).backward()
    loss = ops.cross_entropy(mine(x.to(dev))['out'], masks.to(dev)); loss.backward(); ops.join_side_stream(dev); torch.cuda.synchronize()
    p64, p32 = dict(ref64.named_parameters()), dict(ref.named_parameters())
    names = [k for k, p in mine.named_parameters() if p.grad is not None and p64[k].grad is not None]
    eh = sorted(rel_l2(dict(mine.named_parameters())[k].grad, p64[k].grad) for k in names)[len(names)//2]
    ec = sorted(rel_l2(p32[k].grad, p64[k].grad) for k in names)[len(names)//2]
    rs.append(eh/ec); print(seed, round(eh,4), round(ec,4), round(eh/ec,3), flush=True)
print('ratios', np.round(rs,3), 'median', np.median(rs), 'max', max(rs))
