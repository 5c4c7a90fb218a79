"""FrozenResNetCAM on the HIP path.

Mirrors reference TraditionalModel/ClassificationModel.py:9-41 (identical copy
AlternatingDirectionCutLoss.py:31-63): ResNet-50 with ``replace_stride_with_dilation=[False, False,
True]`` (layer4 dilated, stride-16 features), every trunk parameter frozen, a trainable
``fc: 2048 -> num_classes``, ``forward(x) -> (logits, [f2, f3, f4])`` and the hookable attributes
``layer0 .. layer4, avgpool, fc``.  ImageNet weights cannot be downloaded here: parameters are
randomly initialised as torchvision does; ``load_state_dict`` accepts the reference's checkpoints
(keys ``layer0.0.weight`` ... ``fc.bias``).
"""
import torch
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


def train_fc_only(model, dataloader, device="cuda", epochs=10, lr=1e-3, log=print):
    """Stage 0: reference AlternatingDirectionCutLoss.py:116-141 / ClassificationModel.py:70-106.

    Adam(lr=1e-3) on ``fc`` only with ``nn.CrossEntropyLoss``; ``model.train()`` as in the reference, so the frozen
    trunk's BatchNorm layers normalise with batch statistics and their running statistics keep drifting.
    Per-batch host reads of loss / accuracy are replaced by device accumulators read once per epoch."""
    from .. import ops
    from ..optim import FlatAdam
    model.to(device)
    model.train()
    opt = FlatAdam(list(model.fc.parameters()), lr=lr)
    for epoch in range(epochs):
        tot_loss = torch.zeros((), device=device)
        correct = torch.zeros((), device=device, dtype=torch.long)
        total = 0
        for imgs, (labels, _) in dataloader:
            imgs, labels = imgs.to(device), torch.as_tensor(labels).to(device)
            logits, _ = model(imgs)
            loss = ops.cross_entropy(logits.reshape(logits.shape[0], -1, 1, 1), labels.reshape(-1, 1, 1).long())
            opt.zero_grad()
            loss.backward()
            opt.step()
            tot_loss += loss.detach() * imgs.size(0)
            correct += (logits.detach().argmax(dim=1) == labels).sum()
            total += imgs.size(0)
        if log:
            log(f"Epoch {epoch + 1}/{epochs} - Loss: {tot_loss.item() / total:.4f} - Acc: {100 * correct.item() / total:.2f}%")
    model.eval()
    return model
