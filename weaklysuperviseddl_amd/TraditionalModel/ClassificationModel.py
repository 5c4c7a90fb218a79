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


def _is_device(x):
    return isinstance(x, (str, torch.device, int))


def train_fc_only(model, device="cuda", epochs=10, num_classes=37, *, dataloader=None, val_loader=None, lr=1e-3, log=print):
    """Stage 0: reference ``train_fc_only(model, device, epochs=10, num_classes=37)``
    (ClassificationModel.py:70-106; notebook AlternatingDirectionCutLoss.py:116-141), same positional signature.

    Adam(lr=1e-3) on ``fc`` only with ``nn.CrossEntropyLoss``; ``model.train()`` as in the reference, so the frozen
    trunk's BatchNorm layers normalise with batch statistics and their running statistics keep drifting; the model is
    left in eval mode (:106).  The reference builds its loaders from the Oxford-IIIT Pet download (:75-78, network): here
    they are keyword arguments - ``dataloader`` of ``(imgs, (labels, _))`` batches (required), ``val_loader`` (optional:
    ``evaluate_classification`` after every epoch, :101-104).
    Per-batch host reads of loss / accuracy are replaced by device accumulators read once per epoch."""
    from .. import ops
    from ..optim import FlatAdam
    if not _is_device(device):
        raise TypeError("train_fc_only(model, device, epochs=10, num_classes=37, *, dataloader=...): the second argument is "
                        "the device, as in the reference; the loader is the keyword `dataloader`")
    if dataloader is None:
        raise ValueError("train_fc_only: pass dataloader=... (the reference builds it from the Oxford-IIIT Pet download, "
                         "ClassificationModel.py:75-78, which needs the network)")
    model.to(device)
    opt = FlatAdam(list(model.fc.parameters()), lr=lr)
    for epoch in range(epochs):
        model.train()
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
            log(f"Epoch {epoch + 1}/{epochs} - Train Loss: {tot_loss.item() / total:.4f} - Train Acc: {100 * correct.item() / total:.2f}%")
        if val_loader is not None:
            val_acc, val_f1 = evaluate_classification(model, val_loader, device, num_classes=num_classes, log=None)
            if log:
                log(f"           --> Val Acc: {val_acc:.2f}% - Val F1: {val_f1:.4f}")
    model.eval()
    return model


@torch.no_grad()
def evaluate_classification(model, dataloader, device="cuda", num_classes=37, *, log=print):
    """Reference ``evaluate_classification(model, dataloader, device, num_classes=37)`` (ClassificationModel.py:109-150):
    accuracy (per cent) and macro-F1 over the loader's ``(imgs, (labels, _))`` batches, the reference's formulas
    (precision / recall / F1 with +1e-8 in every denominator, F1 averaged over all ``num_classes`` classes whether
    present or not).  The per-class true/false positive counts come from one confusion-matrix ``bincount`` per batch on
    the device instead of 3 x 37 masked sums, and nothing is read back before the end."""
    model.eval()
    model.to(device)
    conf = torch.zeros(num_classes * num_classes, device=device, dtype=torch.long)
    for imgs, (labels, _) in dataloader:
        imgs, labels = imgs.to(device), torch.as_tensor(labels).to(device).long().view(-1)
        logits, _ = model(imgs)
        preds = logits.argmax(dim=1)
        conf += torch.bincount(labels * num_classes + preds, minlength=num_classes * num_classes)
    conf = conf.view(num_classes, num_classes).double()          # [true, predicted]
    tp = conf.diag()
    fp = conf.sum(0) - tp
    fn = conf.sum(1) - tp
    precision = tp / (tp + fp + 1e-8)
    recall = tp / (tp + fn + 1e-8)
    f1 = 2 * precision * recall / (precision + recall + 1e-8)
    macro_f1 = f1.float().mean().item()
    acc = 100.0 * tp.sum().item() / max(conf.sum().item(), 1.0)
    if log:
        log(f"Evaluation - Accuracy: {acc:.2f}% - F1 Score (macro): {macro_f1:.4f}")
    return acc, macro_f1
