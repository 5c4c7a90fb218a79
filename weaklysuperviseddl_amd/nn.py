"""Parameter-holding modules for the HIP path.

They keep torchvision's attribute / state_dict names (conv1.weight, bn1.running_mean, downsample.0.weight,
classifier.0.convs.1.0.weight ...) so real weights drop in, but never call an ATen compute op: every
forward goes through ``ops`` (libwsdl_hip.so).  ``FusedSequential`` runs Conv2d -> BatchNorm2d [-> ReLU]
triples as one fused node (conv + BN statistics/apply + activation).
"""
import math

import torch
import torch.nn as nn

from . import ops


class Conv2d(nn.Module):
    def __init__(self, cin, cout, k, stride=1, padding=0, dilation=1, bias=False):
        super().__init__()
        self.in_channels, self.out_channels, self.kernel_size = cin, cout, k
        self.stride, self.padding, self.dilation = stride, padding, dilation
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k))
        self.bias = nn.Parameter(torch.empty(cout)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        # nn.Conv2d default init (kaiming_uniform a=sqrt(5)); model builders re-init as torchvision does
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            bound = 1.0 / math.sqrt(self.in_channels * self.kernel_size * self.kernel_size)
            nn.init.uniform_(self.bias, -bound, bound)

    def forward(self, x):
        return ops.conv_bias_act(x, self.weight, self.bias, self.stride, self.padding, self.dilation,
                                 cache=self.__dict__.setdefault("_wsdl_cache", {}))

    def extra_repr(self):
        return (f"{self.in_channels}, {self.out_channels}, k={self.kernel_size}, s={self.stride}, "
                f"p={self.padding}, d={self.dilation}, bias={self.bias is not None}")


class BatchNorm2d(nn.Module):
    def __init__(self, c, eps=1e-5, momentum=0.1):
        super().__init__()
        self.num_features, self.eps, self.momentum = c, eps, momentum
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
        self._pending_steps = 0     # train-mode forwards not yet folded into num_batches_tracked

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        if self._pending_steps:
            self.num_batches_tracked += self._pending_steps
            self._pending_steps = 0
        super()._save_to_state_dict(destination, prefix, keep_vars)

    def _load_from_state_dict(self, *args, **kwargs):
        self._pending_steps = 0
        super()._load_from_state_dict(*args, **kwargs)

    def forward(self, x):
        # the models run BatchNorm fused behind the convolution (FusedSequential / conv_bn); a stand-alone call - a
        # drop-in user doing model.backbone.bn1(x) - takes the same kernels without the convolution
        if self.training:
            self._pending_steps += 1
        return ops.batch_norm(x, self.weight, self.bias, self.running_mean, self.running_var, self.momentum, self.eps,
                              self.training)


class ReLU(nn.Module):
    def forward(self, x):
        # fused into the producing kernel inside the models; stand-alone (hooked / called directly) it is one launch
        return ops.relu(x)


class MaxPool3x3s2(nn.Module):
    def forward(self, x):
        return ops.max_pool_3x3_s2(x)


class GlobalAvgPool(nn.Module):
    """nn.AdaptiveAvgPool2d(1)"""

    def forward(self, x):
        return ops.global_avg_pool(x)


class Dropout(nn.Module):
    def __init__(self, p):
        super().__init__()
        self.p = p
        self.injected_mask = None   # parity tests: uint8 mask used instead of the device RNG
        self._seed = None           # host seed, drawn once; the per-call variation is a device counter (graph-safe)
        self._counter = None

    def forward(self, x):
        if not self.training or self.p == 0.0:
            return x
        if self.injected_mask is not None:
            return ops.dropout(x, self.p, True, mask=self.injected_mask)
        if self._seed is None or self._counter is None or self._counter.device != x.device:
            self._seed = (int(torch.randint(0, 2 ** 62, (1,)).item()) + ops.DROPOUT_SEED_OFFSET[0]) % (1 << 62)
            self._counter = torch.zeros(1, dtype=torch.int64, device=x.device)
        return ops.dropout(x, self.p, True, seed=self._seed, counter=self._counter)


class Linear(nn.Module):
    def __init__(self, fin, fout):
        super().__init__()
        self.in_features, self.out_features = fin, fout
        self.weight = nn.Parameter(torch.empty(fout, fin))
        self.bias = nn.Parameter(torch.empty(fout))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        nn.init.uniform_(self.bias, -1.0 / math.sqrt(fin), 1.0 / math.sqrt(fin))

    def forward(self, x):
        return ops.linear(x, self.weight, self.bias)


def conv_bn(x, conv, bn, relu, residual=None, passthrough=False, link=None, out_holder=None):
    """conv -> bn (batch stats when bn.training, folded running stats otherwise) -> +residual -> relu."""
    if conv.bias is not None:
        raise RuntimeError("conv_bn: a conv followed by BN carries no bias on this path")
    if bn.training:
        bn._pending_steps += 1      # momentum is fixed, so the counter never feeds the arithmetic
    cache = conv.__dict__.setdefault("_wsdl_cache", {})         # derived tensors, keyed on versions / epochs
    return ops.conv_bn_act(x, conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var, conv.stride,
                           conv.padding, conv.dilation, relu, residual, bn.momentum, bn.eps, bn.training, cache,
                           passthrough, link, out_holder if bn.training else None)


class FusedSequential(nn.Sequential):
    """nn.Sequential whose Conv2d, BatchNorm2d[, ReLU] runs execute as single fused nodes."""

    def forward(self, x):
        mods = list(self)
        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, Conv2d) and i + 1 < len(mods) and isinstance(mods[i + 1], BatchNorm2d):
                relu = i + 2 < len(mods) and isinstance(mods[i + 2], ReLU)
                x = conv_bn(x, m, mods[i + 1], relu)
                i += 3 if relu else 2
            else:
                x = m(x)
                i += 1
        return x


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None):
        super().__init__()
        self.conv1 = Conv2d(inplanes, planes, 1)
        self.bn1 = BatchNorm2d(planes)
        self.conv2 = Conv2d(planes, planes, 3, stride=stride, padding=dilation, dilation=dilation)
        self.bn2 = BatchNorm2d(planes)
        self.conv3 = Conv2d(planes, planes * 4, 1)
        self.bn3 = BatchNorm2d(planes * 4)
        self.relu = ReLU()
        self.downsample = downsample

    def forward(self, x):
        # x feeds conv1 AND the identity / downsample branch: route the second use through conv1's node (see
        # ops.conv_bn_act) so that the two input gradients are summed in conv1's dgrad epilogue
        # identity blocks in train mode: the identity branch's gradient goes from the last node to the first through an
        # ops.IdentityLink instead of through a tensor of its own
        train = self.bn1.training and self.bn3.training and torch.is_grad_enabled()
        ds = self.downsample
        proj = (train and ds is not None and len(ds) == 2 and isinstance(ds[0], Conv2d) and isinstance(ds[1], BatchNorm2d)
                and ds[1].training and ds[0].bias is None)
        link = ops.IdentityLink(projection=proj) if (train and (ds is None or proj)) else None
        y, idt = conv_bn(x, self.conv1, self.bn1, True, passthrough=True, link=None if proj else link)
        if proj:
            idt = conv_bn(idt, ds[0], ds[1], False, link=link)       # the projection shortcut, joined to the last node
        elif ds is not None:
            idt = ds(idt)
        y = conv_bn(y, self.conv2, self.bn2, True)
        return conv_bn(y, self.conv3, self.bn3, True, residual=idt, link=link)   # relu(bn3(conv3) + identity)


def make_resnet50_stages(replace_stride_with_dilation):
    """-> (conv1, bn1, layer1..layer4) with torchvision ResNet-50 v1.5 wiring and initialisation."""
    state = {"inplanes": 64, "dilation": 1}

    def stage(planes, blocks, stride, dilate):
        prev = state["dilation"]
        if dilate:
            state["dilation"] *= stride
            stride = 1
        down = None
        if stride != 1 or state["inplanes"] != planes * 4:
            down = FusedSequential(Conv2d(state["inplanes"], planes * 4, 1, stride=stride), BatchNorm2d(planes * 4))
        mods = [Bottleneck(state["inplanes"], planes, stride, prev, down)]
        state["inplanes"] = planes * 4
        for _ in range(1, blocks):
            mods.append(Bottleneck(state["inplanes"], planes, 1, state["dilation"], None))
        return nn.Sequential(*mods)

    conv1 = Conv2d(3, 64, 7, stride=2, padding=3)
    bn1 = BatchNorm2d(64)
    r = replace_stride_with_dilation
    layers = [stage(64, 3, 1, False), stage(128, 4, 2, r[0]), stage(256, 6, 2, r[1]), stage(512, 3, 2, r[2])]
    for m in [conv1] + [mm for l in layers for mm in l.modules()]:
        if isinstance(m, Conv2d):
            nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
    return conv1, bn1, layers
