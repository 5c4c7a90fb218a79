import sys, torch, torch.nn as nn, torch.nn.functional as F
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import oracle
from weaklysuperviseddl_amd.TraditionalModel import build_segmentation_model
from weaklysuperviseddl_amd import ops, nn as wnn
dev = torch.device('cuda:0')
def rel_err(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()
torch.manual_seed(1)
ref = oracle.build_segmentation_model()
mine = build_segmentation_model(); mine.load_state_dict(ref.state_dict())
for m in ref.modules():
    if isinstance(m, nn.Dropout): m.p = 0.0
for m in mine.modules():
    if isinstance(m, wnn.Dropout): m.p = 0.0
mine = mine.to(dev); ref.train(); mine.train()
g = torch.Generator().manual_seed(6)
# head only
feat = torch.randn(4, 2048, 8, 8, generator=g)
fr = feat.clone().requires_grad_(); fm = feat.to(dev).requires_grad_()
o_r = ref.classifier(fr); o_m = mine.classifier(fm)
print('head fwd', rel_err(o_m, o_r))
dy = torch.randn(o_r.shape, generator=g)
o_r.backward(dy); o_m.backward(dy.to(dev))
print('head dfeat', rel_err(fm.grad, fr.grad))
pr = dict(ref.named_parameters())
for k, p in mine.named_parameters():
    if k.startswith('classifier'):
        print(k, '%.2e' % rel_err(p.grad, pr[k].grad))
# ASPP branches individually
for i in range(5):
    fr = feat.clone().requires_grad_(); fm = feat.to(dev).requires_grad_()
    a = ref.classifier[0].convs[i](fr); b = mine.classifier[0].convs[i](fm)
    dyb = torch.randn(a.shape, generator=g)
    a.backward(dyb); b.backward(dyb.to(dev))
    print('branch', i, 'fwd %.2e dx %.2e' % (rel_err(b, a), rel_err(fm.grad, fr.grad)))
# layer4 only
f3 = torch.randn(4, 1024, 8, 8, generator=g)
fr = f3.clone().requires_grad_(); fm = f3.to(dev).requires_grad_()
ref.zero_grad(); mine.zero_grad()
a = ref.backbone.layer4(fr); b = mine.backbone.layer4(fm)
dyb = torch.randn(a.shape, generator=g)
a.backward(dyb); b.backward(dyb.to(dev))
print('layer4 fwd %.2e dx %.2e' % (rel_err(b, a), rel_err(fm.grad, fr.grad)))
for k, p in mine.named_parameters():
    if k.startswith('backbone.layer4'):
        print(k, '%.2e' % rel_err(p.grad, pr[k].grad))
