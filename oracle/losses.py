"""Oracle pairwise-affinity losses (TEST INFRASTRUCTURE ONLY - see oracle/__init__.py).

Restates, for CPU tensors:

  * ``LocalNormalizedCutLoss.forward``        reference AlternatingDirectionCutLoss.py:65-105
  * ``compute_affinities``                    reference AlternatingDirectionCutLoss.py:612-637
  * ``ConstrainToBoundaryLossSingle.forward`` reference AlternatingDirectionBoundaryLoss.py:12-44
    with ``compute_affinities_single`` (:46-70) taken with its *intended* static-call semantics
    (the reference raises TypeError as written).

All three share one neighbourhood core: for the K = w*w-1 offsets (dy major, dx minor, centre
skipped) of a reflect-padded w x w window

    a_k(p) = exp(-|I(p) - I(p+o_k)|^2 / (2 sc^2)  [- |o_k|^2 / (2 ss^2)])
    t_k(p) = sum_c (P_c(p) - P_c(p+o_k))^2

NCut   (normalise=0): softmax inside;  loss = sum_k sum_c mean_{B,H,W}(a_k (dP_c)^2) / (K*C)
Boundary (normalise=1): no softmax;    loss = sum_k mean_{H,W}(a_k t_k) / K      (per image)

Unlike the reference's 24-iteration slicing loop this builds the stacked shifted views once;
values agree to fp32 re-association (checked against golden vectors in tests/).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


def _offsets(window):
    r = window // 2
    return [(dy, dx) for dy in range(-r, r + 1) for dx in range(-r, r + 1) if (dy, dx) != (0, 0)]


def _shifted_stack(t, window):
    """(B,C,H,W) -> (K,B,C,H,W): t evaluated at p+o_k under reflect padding."""
    r = window // 2
    H, W = t.shape[-2:]
    tp = F.pad(t, (r, r, r, r), mode="reflect")
    return torch.stack([tp[..., r + dy:r + dy + H, r + dx:r + dx + W] for dy, dx in _offsets(window)])


def _affinity_stack(image, window, sigma_color, sigma_space):
    """(B,3,H,W) -> (K,B,H,W) affinity maps."""
    d2 = (image.unsqueeze(0) - _shifted_stack(image, window)).pow(2).sum(dim=2)
    expo = -d2 / (2 * sigma_color ** 2)
    if sigma_space is not None and sigma_space > 0:
        sp = torch.tensor([float(dy * dy + dx * dx) for dy, dx in _offsets(window)],
                          dtype=image.dtype).view(-1, 1, 1, 1)
        expo = expo - sp / (2 * sigma_space ** 2)
    return torch.exp(expo)


def compute_affinities(image, sigma_color=0.1, sigma_space=5, window_size=5):
    """Reference AlternatingDirectionCutLoss.py:612-637: list of K (B,1,H,W) maps."""
    a = _affinity_stack(image, window_size, sigma_color, sigma_space)
    return [a[k].unsqueeze(1) for k in range(a.shape[0])]


def pairwise_affinity_loss(preds, image, window=5, sigma_color=0.1, sigma_space=0.0,
                           apply_softmax=True, normalise=0):
    """The generalised kernel contract (include/wsdl_hip.h: wsdl_pairwise_affinity_loss_*).

    preds (B,C,H,W), image (B,3,H,W).  normalise=0 -> scalar; normalise=1 -> (B,) per image.
    """
    P = F.softmax(preds, dim=1) if apply_softmax else preds
    K = window * window - 1
    a = _affinity_stack(image, window, sigma_color, sigma_space)          # (K,B,H,W)
    d = (P.unsqueeze(0) - _shifted_stack(P, window)).pow(2)                # (K,B,C,H,W)
    wd = a.unsqueeze(2) * d
    B, C, H, W = preds.shape
    if normalise == 0:
        return wd.sum() / (B * H * W) / (K * C)
    return wd.sum(dim=(0, 2, 3, 4)) / (H * W) / K


class LocalNormalizedCutLoss(nn.Module):
    """Reference AlternatingDirectionCutLoss.py:65-105 (softmax always applied, no spatial term)."""

    def __init__(self, sigma_color=0.05, window_size=5):
        super().__init__()
        self.sigma_color = sigma_color
        self.window_size = window_size

    def forward(self, preds, images):
        if preds.dim() == 3:
            preds, images = preds.unsqueeze(0), images.unsqueeze(0)
        return pairwise_affinity_loss(preds, images, self.window_size, self.sigma_color, 0.0,
                                      apply_softmax=True, normalise=0)


class ConstrainToBoundaryLossSingle(nn.Module):
    """Reference AlternatingDirectionBoundaryLoss.py:12-70 (input already probabilities)."""

    def __init__(self, sigma_color=0.1, sigma_space=5, window_size=5, eps=1e-8):
        super().__init__()
        self.sigma_color, self.sigma_space = sigma_color, sigma_space
        self.window_size, self.eps = window_size, eps

    def forward(self, preds, image):
        out = pairwise_affinity_loss(preds.unsqueeze(0), image.unsqueeze(0), self.window_size,
                                     self.sigma_color, self.sigma_space, apply_softmax=False,
                                     normalise=1)
        return out[0]

    @staticmethod
    def compute_affinities_single(image, sigma_color=0.1, sigma_space=5, window_size=5):
        a = _affinity_stack(image.unsqueeze(0), window_size, sigma_color, sigma_space)
        return [a[k] for k in range(a.shape[0])]                           # K x (1,H,W)


# ---------------------------------------------------------------------------------------------------------------
# Lovasz-softmax (reference TraditionalModel/LossFunctions/Lovasz-Softmax_Loss.py; optional loss of
# train_segmentation_model, SegmentationModel.py:103-105: ``lovasz_softmax(F.softmax(outputs, 1), masks,
# classes='present', per_image=False, ignore=None)``).  Restated from the algorithm (Berman et al., Alg. 1):
def lovasz_grad(gt_sorted):
    """Gradient of the Lovasz extension of the Jaccard loss w.r.t. sorted errors (:11-23): with G = sum(gt),
    J_k = 1 - (G - cumsum(gt)_k) / (G + cumsum(1 - gt)_k), the k-th entry is J_k - J_{k-1} (J_{-1} = 0)."""
    gt = gt_sorted.float()
    total = gt.sum()
    inter = total - gt.cumsum(0)
    union = total + (1.0 - gt).cumsum(0)
    jac = 1.0 - inter / union
    out = jac.clone()
    out[1:] = jac[1:] - jac[:-1]
    return out


def lovasz_softmax_flat(probas, labels, classes="present"):
    """probas (P, C), labels (P,) -> mean over the (present) classes of <sorted |fg - p_c|, lovasz_grad(fg sorted)>
    (:164-192).  The gradient vector is a constant of the sort order (not differentiated through)."""
    if probas.numel() == 0:
        return probas * 0.0
    C = probas.shape[1]
    losses = []
    which = range(C) if classes in ("all", "present") else classes
    for c in which:
        fg = (labels == c).float()
        if classes == "present" and fg.sum() == 0:
            continue
        errors = (fg - probas[:, c]).abs()
        errors_sorted, perm = torch.sort(errors, 0, descending=True)
        losses.append(torch.dot(errors_sorted, lovasz_grad(fg[perm]).detach()))
    if not losses:
        return probas.sum() * 0.0            # the reference's mean([]) = 0
    return sum(losses) / len(losses)


def lovasz_softmax(probas, labels, classes="present", per_image=False, ignore=None):
    """probas (B, C, H, W) class probabilities, labels (B, H, W) (:146-161, flatten_probas :195-211)."""
    def flat(p, l):
        Cc = p.shape[1]
        p = p.permute(0, 2, 3, 1).reshape(-1, Cc)
        l = l.reshape(-1)
        if ignore is None:
            return p, l
        keep = l != ignore
        return p[keep], l[keep]
    if per_image:
        vals = [lovasz_softmax_flat(*flat(p.unsqueeze(0), l.unsqueeze(0)), classes=classes) for p, l in zip(probas, labels)]
        return sum(vals) / len(vals)
    return lovasz_softmax_flat(*flat(probas, labels), classes=classes)
