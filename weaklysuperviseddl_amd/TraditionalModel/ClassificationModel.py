"""FrozenResNetCAM on the HIP path.

Mirrors reference TraditionalModel/ClassificationModel.py:9-41 (identical copy
AlternatingDirectionCutLoss.py:31-63): ResNet-50 with ``replace_stride_with_dilation=[False, False,
True]`` (layer4 dilated, stride-16 features), every trunk parameter frozen, a trainable
``fc: 2048 -> num_classes``, ``forward(x) -> (logits, [f2, f3, f4])`` and the hookable attributes
``layer0 .. layer4, avgpool, fc``.  ImageNet weights cannot be downloaded here: parameters are
randomly initialised as torchvision does; ``load_state_dict`` accepts the reference's checkpoints
(keys ``layer0.0.weight`` ... ``fc.bias``).
"""
import torch.nn as nn

from .. import nn as wnn


class FrozenResNetCAM(nn.Module):
    def __init__(self, num_classes=37):
        super().__init__()
        conv1, bn1, (l1, l2, l3, l4) = wnn.make_resnet50_stages((False, False, True))
        self.layer0 = wnn.FusedSequential(conv1, bn1, wnn.ReLU(), wnn.MaxPool3x3s2())
        self.layer1, self.layer2, self.layer3, self.layer4 = l1, l2, l3, l4
        for p in self.parameters():
            p.requires_grad = False
        self.avgpool = wnn.GlobalAvgPool()
        self.fc = wnn.Linear(2048, num_classes)

    def forward(self, x):
        x = self.layer0(x)
        f1 = self.layer1(x)
        f2 = self.layer2(f1)
        f3 = self.layer3(f2)
        f4 = self.layer4(f3)
        logits = self.fc(self.avgpool(f4).flatten(1))
        return logits, [f2, f3, f4]
