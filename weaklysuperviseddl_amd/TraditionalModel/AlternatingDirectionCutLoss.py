"""Normalised-cut loss, compute_affinities, refine_pseudo_mask and train_model on the HIP path.

Mirrors reference TraditionalModel/AlternatingDirectionCutLoss.py:
  * ``LocalNormalizedCutLoss``  :65-105  - softmax inside, reflect-padded w x w window, colour-only
    Gaussian affinity, ``sum_k sum_c mean(a_k (P_c - P_c')^2) / (K*C)``; forward AND gradient are one
    fused kernel launch (the reference issues ~24*(6+4C) full-tensor ops and as many autograd nodes);
  * ``compute_affinities``      :612-637 - list of K (B,1,H,W) colour+spatial affinity maps;
  * ``refine_pseudo_mask``      :709-767 - Adam on a free tensor X: KL(softmax X || S) + lam_dyn*NCut
    with the dynamic weight kept ON DEVICE (the reference does two ``.item()`` syncs per step);
    softmax is applied twice to X on the NCut branch, as the reference does;
  * ``train_model``             :684-707 - CE-only training epochs.
"""
import torch
import torch.nn as nn

from .. import ops
from ..optim import FlatAdam
from .SegmentationModel import train_step


class LocalNormalizedCutLoss(nn.Module):
    def __init__(self, sigma_color=0.05, window_size=5):
        super().__init__()
        self.sigma_color = sigma_color
        self.window_size = window_size

    def forward(self, preds, images):
        if preds.dim() == 3:
            preds, images = preds.unsqueeze(0), images.unsqueeze(0)
        return ops.pairwise_affinity_loss(preds, images, self.window_size, self.sigma_color, 0.0,
                                          apply_softmax=True, normalise=0)


def compute_affinities(image, sigma_color=0.1, sigma_space=5, window_size=5):
    a = ops.compute_affinities(image, sigma_color, sigma_space, window_size)      # (K,B,1,H,W)
    return [a[k] for k in range(a.shape[0])]


def refine_pseudo_mask(model, image, mask, lambda_boundary=0.1, threshold=0.5, lr=1e-2, num_steps=20,
                       sigma_color=0.1, window_size=5):
    device = next(model.parameters()).device
    image = image.to(device)
    model.eval()
    x = image.unsqueeze(0)
    with torch.no_grad():
        S = ops.softmax_channels(model(x)["out"])
    onehot = torch.stack([(mask != 255), (mask == 255)]).to(device=device, dtype=torch.float32)   # one_hot(mask==255)
    X = onehot.unsqueeze(0).contiguous().requires_grad_(True)
    opt = FlatAdam([X], lr=lr)
    ncut = LocalNormalizedCutLoss(sigma_color=sigma_color, window_size=window_size)
    for _ in range(num_steps):
        opt.zero_grad()
        Xn = ops.softmax_channels(X)
        kl = ops.kl_div_batchmean(Xn, S)
        nc = ncut(Xn[0], x[0])
        lam = (lambda_boundary * (kl.detach() / (nc.detach() + 1e-6)))     # stays on the device
        loss = kl + lam * nc
        loss.backward()
        opt.step()
    with torch.no_grad():
        Xf = ops.softmax_channels(X)
    return (Xf[0, 1] > threshold).float()


def train_model(model, optimizer, train_loader, num_epochs=3, device="cuda", log=print):
    """CE-only epochs (the reference's ``train_model``; criterion is nn.CrossEntropyLoss())."""
    model.train()
    for epoch in range(num_epochs):
        total = torch.zeros((), device=device)
        for batch in train_loader:
            images, masks = batch[0].to(device), batch[1].to(device)
            if images.size(0) == 1:       # SegmentationModel.py:97-98: BN cannot normalise one pooled value
                continue
            total += train_step(model, optimizer, images, masks)
        if log:
            log(f"Epoch {epoch + 1}/{num_epochs}, Loss: {total.item():.4f}")
    return model
