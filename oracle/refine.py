"""Oracle pseudo-mask refinement (TEST INFRASTRUCTURE ONLY - see oracle/__init__.py).

Restates ``refine_pseudo_mask`` - reference TraditionalModel/AlternatingDirectionCutLoss.py:709-767:
one eval-mode forward S = softmax(model(img)['out']); free variable X = one_hot(mask == 255);
``num_steps`` Adam steps on  KL(softmax(X) || S) [batchmean, log(X+1e-8)] + lam_dyn * NCut
where NCut receives the already-softmaxed X (softmax applied twice) and
lam_dyn = lam * kl / (ncut + 1e-6) is a detached scalar; returns softmax(X)[0,1] > threshold.
"""
import torch
import torch.nn.functional as F

from .losses import LocalNormalizedCutLoss


def refine_pseudo_mask(model, image, mask, lambda_boundary=0.1, threshold=0.5, lr=1e-2,
                       num_steps=20, sigma_color=0.1, window_size=5, return_trace=False):
    model.eval()
    x = image.unsqueeze(0)
    with torch.no_grad():
        S = F.softmax(model(x)["out"], dim=1)
    onehot = F.one_hot((mask == 255).long(), num_classes=2).permute(2, 0, 1).float()
    X = onehot.unsqueeze(0).clone().requires_grad_(True)
    opt = torch.optim.Adam([X], lr=lr)
    ncut = LocalNormalizedCutLoss(sigma_color=sigma_color, window_size=window_size)
    trace = []
    for _ in range(num_steps):
        opt.zero_grad()
        Xn = F.softmax(X, dim=1)
        kl = F.kl_div((Xn + 1e-8).log(), S, reduction="batchmean")
        nc = ncut(Xn[0], x[0])
        lam = lambda_boundary * (kl.item() / (nc.item() + 1e-6))
        loss = kl + lam * nc
        loss.backward()
        opt.step()
        trace.append((kl.item(), nc.item(), loss.item()))
    out = (F.softmax(X, dim=1)[0, 1] > threshold).float().detach()
    return (out, trace) if return_trace else out
