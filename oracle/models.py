"""Oracle models (TEST INFRASTRUCTURE ONLY - see oracle/__init__.py).

PyTorch-CPU restatement of the two networks on the hot path.  The reference
does not contain them: it calls torchvision (unpinned, not vendored, absent
from this image):

  * ``torchvision.models.resnet50(pretrained=True,
    replace_stride_with_dilation=[False, False, True])`` wrapped by
    ``FrozenResNetCAM``   - reference TraditionalModel/ClassificationModel.py:9-41
    (identical copy AlternatingDirectionCutLoss.py:31-63);
  * ``torchvision.models.segmentation.deeplabv3_resnet50(pretrained=True)`` with
    ``classifier[4] = nn.Conv2d(256, 2, 1)``
                          - reference TraditionalModel/SegmentationModel.py:85-88,
    AlternatingDirectionCutLoss.py:784-787, FullySupervisedModel/SupervisedModel.py:13-16.

The architecture restated here is the published torchvision one (ResNet v1.5
bottlenecks, stride on the 3x3; DeepLabHead = ASPP(12,24,36) + 3x3 + 1x1;
FCNHead aux classifier).  ``state_dict`` keys are torchvision's, so real
weights would drop in.  Pinned by parameter-count / key-set self checks only.
"""
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F


class Bottleneck(nn.Module):
    """1x1 -> 3x3(stride, dilation) -> 1x1(x4), residual, ReLU after the add."""

    expansion = 4

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=dilation,
                               dilation=dilation, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=False)
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        return self.relu(y + idt)


class ResNet50Trunk(nn.Module):
    """ResNet-50 with torchvision's attribute names (conv1, bn1, layer1..4, fc)."""

    def __init__(self, replace_stride_with_dilation=(False, False, False), num_classes=1000,
                 with_fc=True):
        super().__init__()
        self.inplanes = 64
        self.dilation = 1
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=False)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        self.layer1 = self._stage(64, 3, 1, False)
        self.layer2 = self._stage(128, 4, 2, replace_stride_with_dilation[0])
        self.layer3 = self._stage(256, 6, 2, replace_stride_with_dilation[1])
        self.layer4 = self._stage(512, 3, 2, replace_stride_with_dilation[2])
        if with_fc:
            self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
            self.fc = nn.Linear(2048, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1.0)
                nn.init.constant_(m.bias, 0.0)

    def _stage(self, planes, blocks, stride, dilate):
        prev_dil = self.dilation
        if dilate:
            self.dilation *= stride
            stride = 1
        down = None
        if stride != 1 or self.inplanes != planes * 4:
            down = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, 1, stride=stride, bias=False),
                                 nn.BatchNorm2d(planes * 4))
        mods = [Bottleneck(self.inplanes, planes, stride, prev_dil, down)]
        self.inplanes = planes * 4
        for _ in range(1, blocks):
            mods.append(Bottleneck(self.inplanes, planes, 1, self.dilation, None))
        return nn.Sequential(*mods)


class FrozenResNetCAM(nn.Module):
    """Reference ClassificationModel.py:9-41: frozen trunk, layer0..layer4, fc 2048->nc."""

    def __init__(self, num_classes=37):
        super().__init__()
        trunk = ResNet50Trunk(replace_stride_with_dilation=(False, False, True))
        for p in trunk.parameters():
            p.requires_grad = False
        self.layer0 = nn.Sequential(trunk.conv1, trunk.bn1, trunk.relu, trunk.maxpool)
        self.layer1, self.layer2 = trunk.layer1, trunk.layer2
        self.layer3, self.layer4 = trunk.layer3, trunk.layer4
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(2048, num_classes)

    def forward(self, x):
        x = self.layer0(x)
        f1 = self.layer1(x)
        f2 = self.layer2(f1)
        f3 = self.layer3(f2)
        f4 = self.layer4(f3)
        logits = self.fc(self.avgpool(f4).flatten(1))
        return logits, [f2, f3, f4]


def _conv_bn_relu(cin, cout, k, dilation=1):
    pad = 0 if k == 1 else dilation
    return [nn.Conv2d(cin, cout, k, padding=pad, dilation=dilation, bias=False),
            nn.BatchNorm2d(cout), nn.ReLU()]


class _ASPPPool(nn.Sequential):
    def __init__(self, cin, cout):
        super().__init__(nn.AdaptiveAvgPool2d(1), nn.Conv2d(cin, cout, 1, bias=False),
                         nn.BatchNorm2d(cout), nn.ReLU())

    def forward(self, x):
        size = x.shape[-2:]
        for m in self:
            x = m(x)
        return F.interpolate(x, size=size, mode="bilinear", align_corners=False)


class ASPP(nn.Module):
    def __init__(self, cin=2048, rates=(12, 24, 36), cout=256):
        super().__init__()
        branches = [nn.Sequential(*_conv_bn_relu(cin, cout, 1))]
        branches += [nn.Sequential(*_conv_bn_relu(cin, cout, 3, r)) for r in rates]
        branches.append(_ASPPPool(cin, cout))
        self.convs = nn.ModuleList(branches)
        self.project = nn.Sequential(*_conv_bn_relu(len(branches) * cout, cout, 1), nn.Dropout(0.5))

    def forward(self, x):
        return self.project(torch.cat([b(x) for b in self.convs], dim=1))


class _Backbone(nn.Module):
    """IntermediateLayerGetter equivalent: returns {'out': layer4, 'aux': layer3}."""

    def __init__(self):
        super().__init__()
        t = ResNet50Trunk(replace_stride_with_dilation=(False, True, True), with_fc=False)
        self.conv1, self.bn1, self.relu, self.maxpool = t.conv1, t.bn1, t.relu, t.maxpool
        self.layer1, self.layer2, self.layer3, self.layer4 = t.layer1, t.layer2, t.layer3, t.layer4

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer2(self.layer1(x))
        f3 = self.layer3(x)
        return OrderedDict(out=self.layer4(f3), aux=f3)


class DeepLabV3ResNet50(nn.Module):
    """torchvision ``deeplabv3_resnet50`` layout: backbone / classifier / aux_classifier."""

    def __init__(self, num_classes=21, aux_loss=True):
        super().__init__()
        self.backbone = _Backbone()
        self.classifier = nn.Sequential(ASPP(), *_conv_bn_relu(256, 256, 3, 1),
                                        nn.Conv2d(256, num_classes, 1))
        self.aux_classifier = None
        if aux_loss:
            self.aux_classifier = nn.Sequential(*_conv_bn_relu(1024, 256, 3, 1), nn.Dropout(0.1),
                                                nn.Conv2d(256, num_classes, 1))

    def forward(self, x):
        size = x.shape[-2:]
        feats = self.backbone(x)
        res = OrderedDict()
        res["out"] = F.interpolate(self.classifier(feats["out"]), size=size, mode="bilinear",
                                   align_corners=False)
        if self.aux_classifier is not None:
            res["aux"] = F.interpolate(self.aux_classifier(feats["aux"]), size=size,
                                       mode="bilinear", align_corners=False)
        return res


def build_segmentation_model(num_classes=2, aux_loss=True):
    """Reference SegmentationModel.py:85-88: 21-class model, then classifier[4] swapped."""
    m = DeepLabV3ResNet50(num_classes=21, aux_loss=aux_loss)
    m.classifier[4] = nn.Conv2d(256, num_classes, 1)
    return m
