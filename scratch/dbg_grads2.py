import sys, torch, torch.nn as nn, torch.nn.functional as F
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import oracle
from weaklysuperviseddl_amd.TraditionalModel import build_segmentation_model
from weaklysuperviseddl_amd import ops, nn as wnn
from test_hip_models import randomise_bn
dev = torch.device('cuda:0')
def rel_err(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()
for rnd in (False, True):
    torch.manual_seed(1)
    ref = oracle.build_segmentation_model()
    if rnd: randomise_bn(ref, 2)
    mine = build_segmentation_model(); mine.load_state_dict(ref.state_dict())
    for m in ref.modules():
        if isinstance(m, nn.Dropout): m.p = 0.0
    for m in mine.modules():
        if isinstance(m, wnn.Dropout): m.p = 0.0
    mine = mine.to(dev); ref.train(); mine.train()
    g = torch.Generator().manual_seed(6)
    x = torch.randn(4, 3, 64, 64, generator=g)
    masks = (torch.rand(4, 64, 64, generator=g) > 0.5).long()
    grads_r, grads_m = {}, {}
    def keep(d, name):
        def h(gr): d[name] = gr.detach().cpu()
        return h
    fr = ref.backbone(x); fr['out'].register_hook(keep(grads_r, 'f4')); fr['aux'].register_hook(keep(grads_r, 'f3'))
    hr = ref.classifier(fr['out']); hr.register_hook(keep(grads_r, 'head'))
    out_r = F.interpolate(hr, size=(64, 64), mode='bilinear', align_corners=False)
    loss_r = F.cross_entropy(out_r, masks); loss_r.backward()
    fm = mine.backbone(x.to(dev)); fm['out'].register_hook(keep(grads_m, 'f4')); fm['aux'].register_hook(keep(grads_m, 'f3'))
    hm = mine.classifier(fm['out']); hm.register_hook(keep(grads_m, 'head'))
    out_m = ops.bilinear_resize(hm, (64, 64))
    loss_m = ops.cross_entropy(out_m, masks.to(dev)); loss_m.backward()
    print('randomise', rnd, 'loss', loss_r.item(), loss_m.item(), 'out', rel_err(out_m, out_r), 'f4', rel_err(fm['out'], fr['out']))
    for k in ('head', 'f4', 'f3'):
        print('  grad', k, '%.3e' % rel_err(grads_m[k], grads_r[k]), 'absmax %.3e' % grads_r[k].abs().max().item())
    pr = dict(ref.named_parameters())
    items = [(k, rel_err(p.grad, pr[k].grad)) for k, p in mine.named_parameters() if p.grad is not None]
    for k, v in items[::-1][:40]:
        print('  %-45s %.2e  |g| %.2e' % (k, v, pr[k].grad.abs().max().item()))
