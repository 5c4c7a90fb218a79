"""Normalised-cut loss, compute_affinities, refine_pseudo_mask and train_model on the HIP path.

Mirrors reference TraditionalModel/AlternatingDirectionCutLoss.py:
  * ``LocalNormalizedCutLoss``  :65-105  - softmax inside, reflect-padded w x w window, colour-only
    Gaussian affinity, ``sum_k sum_c mean(a_k (P_c - P_c')^2) / (K*C)``; forward AND gradient are one
    fused kernel launch (the reference issues ~24*(6+4C) full-tensor ops and as many autograd nodes);
  * ``compute_affinities``      :612-637 - list of K (B,1,H,W) colour+spatial affinity maps;
  * ``refine_pseudo_mask``      :709-767 - Adam on a free tensor X: KL(softmax X || S) + lam_dyn*NCut
    with the dynamic weight kept ON DEVICE (the reference does two ``.item()`` syncs per step);
    softmax is applied twice to X on the NCut branch, as the reference does;
  * ``train_model``             :684-707 - CE-only training epochs;
  * ``run_alternating_training`` :791-818 - the outer loop: train <-> five chained refinement passes, in memory.
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


def network_soft_prediction(model, images):
    """S = softmax(model(images)['out']) in eval mode (AlternatingDirectionCutLoss.py:715-720): the constant target of
    the KL term.  It does not depend on the mask, so the chained refinement repeats of one alternation share it."""
    model.eval()
    with torch.no_grad():
        return ops.softmax_channels(model(images)["out"]).contiguous()


def refine_pseudo_masks_batched(model, images, masks, lambda_boundary=0.1, threshold=0.5, lr=1e-2, num_steps=20,
                                sigma_color=0.1, window_size=5, S=None, affinity_cache=None):
    """``refine_pseudo_mask`` for N images at once: images (N,3,H,W), masks (N,H,W) with
    foreground = 255 -> (N,H,W) float masks.  Same arithmetic per image as the reference's per-image loop
    (AlternatingDirectionCutLoss.py:803-810); each step is six kernel launches for the whole batch and no host
    synchronisation (the reference: ~2 x 24 x 14 tiny kernels and two ``.item()`` syncs per image and step).
    ``S``: the network's soft prediction if the caller already has it (``network_soft_prediction``);
    ``affinity_cache``: ``ops.pairwise_cache(images, window_size, sigma_color)`` if the caller already has it.  The image
    is constant over the steps, so its colour affinities are computed once per call (once per chunk in
    ``refine_dataset``) instead of 24 exponentials per pixel and step - bit-identical results."""
    import ctypes as C
    from .._lib import lib, check
    device = next(model.parameters()).device
    images = images.to(device).contiguous()
    N, _, H, W = images.shape
    if S is None:
        S = network_soft_prediction(model, images)
    if affinity_cache is None:
        affinity_cache = ops.pairwise_cache(images, window_size, sigma_color)
    with torch.no_grad():
        fg = (masks.to(device) == 255)
        X = torch.stack([~fg, fg], dim=1).to(torch.float32).contiguous()
        Xn, dkl, dnc, dXn, dX = (torch.empty_like(X) for _ in range(5))
        m, v = torch.zeros_like(X), torch.zeros_like(X)
        kl = torch.empty(N, device=device)
        nc = torch.empty(N, device=device)
        per = 2 * H * W
        ws = ops.workspace(max(lib().wsdl_pairwise_workspace(N, H, W), N * 64 * 4, lib().wsdl_reduce_workspace()), device)
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        p = lambda t: C.c_void_p(t.data_ptr())
        for step in range(1, num_steps + 1):
            check(lib().wsdl_softmax_fwd(p(X), p(Xn), N, 2, H * W, st))
            check(lib().wsdl_kl_div_per_image_fwd_bwd(p(Xn), p(S), p(kl), p(dkl), N, per, p(ws), ws.numel(), st))
            # NCut on the already-softmaxed X (softmax applied again inside, as the reference does - D8);
            # normalise=1 gives sum/(H*W*K) per image, the reference divides by K*C: nc_scale = 1/C
            check(lib().wsdl_pairwise_affinity_loss_fwd_bwd(p(Xn), p(images), p(nc), p(dnc), N, 2, H, W, int(window_size),
                                                            float(sigma_color), 0.0, 1, 1, p(affinity_cache), p(ws),
                                                            ws.numel(), st))
            check(lib().wsdl_refine_combine(p(dkl), p(dnc), p(kl), p(nc), float(lambda_boundary), 0.5, p(dXn), N, per, st))
            check(lib().wsdl_softmax_bwd(p(Xn), p(dXn), p(dX), N, 2, H * W, st))
            ops.adam_step_flat(X.view(-1), dX.view(-1), m.view(-1), v.view(-1), lr, 0.9, 0.999, 1e-8, step)
        check(lib().wsdl_softmax_fwd(p(X), p(Xn), N, 2, H * W, st))
        return (Xn[:, 1] > threshold).float()


train_loader = None      # the reference's ``train_model`` reads a module-level ``train_loader`` (AlternatingDirectionCutLoss.py:692,781)


def train_model(model, optimizer, criterion_ce=None, num_epochs=3, *, train_loader=None, device="cuda", log=print,
                max_steps_per_epoch=None):
    """Reference ``train_model(model, optimizer, criterion_ce, num_epochs=3)`` (AlternatingDirectionCutLoss.py:684-707),
    same positional signature: CE-only epochs over ``train_loader``.

    ``criterion_ce``: the reference passes ``nn.CrossEntropyLoss()`` (:789) - mapped onto the fused HIP
    softmax-cross-entropy kernel (``SegmentationModel.resolve_criterion``); None means the same; any other callable is
    applied to ``(outputs, masks)``.  The loader: keyword ``train_loader`` - a DataLoader of (images, masks[, names]) or a
    callable returning a fresh iterator per epoch (``InMemoryPseudoDataset.batches``) - or, as in the reference, the
    module-level ``train_loader`` of this module.  For callers of the earlier form ``train_model(model, optimizer, loader,
    ...)`` an ITERABLE in the third position (a DataLoader, a list of batches) is still taken as the loader.
    Returns the per-epoch summed losses (the number the reference prints)."""
    if train_loader is None and not isinstance(criterion_ce, nn.Module) and \
            (hasattr(criterion_ce, "__iter__") or hasattr(criterion_ce, "__len__")):
        train_loader, criterion_ce = criterion_ce, None           # earlier call form: an iterable loader in third position
    if train_loader is None:
        train_loader = globals().get("train_loader")
    if train_loader is None:
        raise ValueError("train_model: no data - pass train_loader=... or set "
                         "weaklysuperviseddl_amd.TraditionalModel.AlternatingDirectionCutLoss.train_loader "
                         "(the reference reads a module-level train_loader, AlternatingDirectionCutLoss.py:692)")
    from .SegmentationModel import resolve_criterion
    criterion = None if criterion_ce is None else resolve_criterion(criterion_ce)
    model.train()
    totals = []
    for epoch in range(num_epochs):
        total = torch.zeros((), device=device)
        it = train_loader() if callable(train_loader) else train_loader
        for step, batch in enumerate(it):
            if max_steps_per_epoch is not None and step >= max_steps_per_epoch:
                break
            images, masks = batch[0].to(device), batch[1].to(device)
            if images.size(0) == 1:       # SegmentationModel.py:97-98: BN cannot normalise one pooled value
                continue
            total += train_step(model, optimizer, images, masks, criterion=criterion)
        totals.append(total)
        if log:
            log(f"Epoch {epoch + 1}/{num_epochs}, Loss: {total.item():.4f}")
    return totals


def refine_dataset(model, dataset, repeats=5, chunk=64, threshold=0.3, lr=1e-4, num_steps=10, lambda_boundary=0.1,
                   sigma_color=0.1, window_size=5):
    """Step 2 of an alternation (AlternatingDirectionCutLoss.py:803-810): ``repeats`` passes over the data set, each
    refining every pseudo mask and overwriting it - pass r+1 starts from pass r's thresholded masks, because the
    reference's dataset re-reads the PNG it has just overwritten.  In memory: ``dataset.set_masks``.  The network is
    frozen during the passes, so its soft prediction is computed once per chunk, not once per pass and image."""
    n = len(dataset)
    for s in range(0, n, chunk):
        idx = torch.arange(s, min(s + chunk, n), device=dataset.images.device)
        imgs = dataset.images[idx]
        S = network_soft_prediction(model, imgs)
        cache = ops.pairwise_cache(imgs, window_size, sigma_color)
        for _ in range(repeats):
            refined = refine_pseudo_masks_batched(model, imgs, dataset.masks[idx], lambda_boundary=lambda_boundary,
                                                  threshold=threshold, lr=lr, num_steps=num_steps,
                                                  sigma_color=sigma_color, window_size=window_size, S=S,
                                                  affinity_cache=cache)
            dataset.set_masks(idx, refined)
    return dataset


def run_alternating_training(model, optimizer, dataset, num_alternations=10, epochs_per_round=10, refine_repeats=5,
                             first_batch_size=4, later_batch_size=32, refine_chunk=64, refine_kwargs=None,
                             evaluate=None, seed=0, device="cuda", log=print):
    """The alternating-direction outer loop (reference AlternatingDirectionCutLoss.py:791-818; the modular re-write
    AlternatingDirectionBoundaryLoss.py:153-206 is dead code): per alternation
      1. ``train_model`` for ``epochs_per_round`` epochs on the current pseudo masks (batch 4 in the first
         alternation, 32 afterwards - :781, :818),
      2. optional ``evaluate(model)`` (:795),
      3. ``refine_dataset``: five chained refinement passes over every pseudo mask (:803-810, threshold 0.3, lr 1e-4,
         10 steps, lambda 0.1), masks overwritten in place, data set "re-opened" (:813-818).
    ``dataset`` is THIS RANK's ``InMemoryPseudoDataset`` shard.  Under data parallelism (``torch.distributed``
    initialised, a ``dp.GradBucketReducer`` on the optimizer) every rank must run the same number of optimiser steps:
    the per-epoch step count is the minimum over the ranks; refinement is per-image independent and needs no
    collective."""
    import torch.distributed as dist
    rk = dict(threshold=0.3, lr=1e-4, num_steps=10, lambda_boundary=0.1)
    rk.update(refine_kwargs or {})
    dp = dist.is_initialized() and dist.get_world_size() > 1
    rank = dist.get_rank() if dp else 0
    gen = torch.Generator().manual_seed(seed * 1000003 + rank)
    history = []
    for it in range(num_alternations):
        bs = first_batch_size if it == 0 else later_batch_size
        steps = dataset.num_batches(bs)
        if dp:
            from ..dp import control_group
            t = torch.tensor([steps], dtype=torch.int64)
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=control_group())
            steps = int(t.item())
        losses = train_model(model, optimizer, None, epochs_per_round,
                             train_loader=lambda: dataset.batches(bs, shuffle=True, generator=gen, limit=steps),
                             device=device, log=log if rank == 0 else None)
        metrics = evaluate(model) if evaluate is not None else None
        if log and rank == 0 and metrics is not None:
            log(f"Iteration {it + 1}: Evaluation -> {metrics}")
        refine_dataset(model, dataset, repeats=refine_repeats, chunk=refine_chunk, **rk)
        history.append({"losses": losses, "metrics": metrics})
    if log and rank == 0:
        log("Alternating training and pseudo mask updates completed.")
    return history
