"""Pseudo-mask generation on the HIP path.

Mirrors reference TraditionalModel/PsuedoMasks.py: ``keep_largest`` (:15-21) and
``generate_pseudo_masks`` (:23-79).  Differences, all outside the arithmetic:
  * CAM + threshold run batched on the device (``LayerCAMGenerator.generate_batch`` with the
    threshold fused into the epilogue kernel) - one device->host copy of uint8 masks per batch instead
    of a float map per image;
  * output directories are parameters (the reference hard-codes /content/...); ``write_png=False``
    keeps the masks in memory for the in-memory hand-off to stage 2
    (``generate_pseudo_masks.last_masks`` / ``.last_ids`` / ``.last_images``; ``stage_handoff`` turns them into
    what ``PseudoSegmentationDataset`` would have read back from the PNGs);
  * ``rank`` / ``world``: stage 1 is per-image independent (eval-mode BN), so under data parallelism the loader's
    batches are dealt round-robin to the ranks - batch j belongs to rank j % world - with NO collective
    (SURVEY.md 8e); image ids stay the global ones, so the union over the ranks is the single-process result;
  * ``streams``: that many of the loader's batches are in flight on the device at once (``LayerCAMGenerator.generate_batches``);
    the masks are the same bit for bit;
  * ``keep_largest(mask)`` is the reference's host function (skimage there; scipy.ndimage here - 8-connectivity, raster
    label order, first label wins area ties, empty mask returned unchanged).  ``generate_pseudo_masks`` itself labels
    the whole batch on the device (``keep_largest_batched`` -> ``ops.keep_largest_batched``, one workgroup per mask, the
    same result bit for bit): one device->host copy of the FINAL masks per batch, and none at all with
    ``keep_on_device=True`` - the in-memory hand-off to stage 2 then never synchronises with the host.
"""
import os

import numpy as np
import torch
from scipy import ndimage

from .. import ops

_EIGHT = np.ones((3, 3), dtype=bool)


def keep_largest(mask):
    lab, n = ndimage.label(np.asarray(mask) != 0, structure=_EIGHT)
    if n == 0:
        return mask
    areas = np.bincount(lab.ravel(), minlength=n + 1)[1:]
    return (lab == (int(np.argmax(areas)) + 1)).astype(np.uint8)


def _to_device_async(t, device):
    """Host tensors go through pinned memory: a copy from pageable memory makes the host wait until the stream has
    drained - one such copy per batch (the labels) serialised stage 1 with whatever the device was still running."""
    if t.is_cuda or torch.device(device).type != "cuda":
        return t.to(device)
    return t.pin_memory().to(device, non_blocking=True)


def keep_largest_batched(masks):
    """``keep_largest`` for a (N,H,W) batch of uint8 device masks, on the device."""
    return ops.keep_largest_batched(masks)


def _to_png_u8(t):
    """torchvision.utils.save_image's quantisation: mul(255).add_(0.5).clamp_(0, 255) -> uint8, HWC."""
    return t.mul(255).add(0.5).clamp(0, 255).permute(1, 2, 0).to(torch.uint8).cpu().numpy()


def generate_pseudo_masks(loader, layercam_gen, cam_thresh=0.3, alpha=1.0, keep_largest_masks=True,
                          run_id="default", out_root="/content", max_images=500, write_png=True,
                          device="cuda", rank=0, world=1, keep_images=False, streams=3, keep_on_device=False,
                          device_batch=0):
    mask_dir = os.path.join(out_root, f"pseudo_masks_{run_id}")
    image_dir = os.path.join(out_root, f"images_{run_id}")
    if write_png:
        from PIL import Image
        for d in (mask_dir, image_dir):
            os.makedirs(d, exist_ok=True)
            if rank == 0:
                for f in os.listdir(d):
                    os.remove(os.path.join(d, f))
        if world > 1:
            import torch.distributed as dist
            if dist.is_initialized():
                dist.barrier()                  # nobody writes before rank 0 has emptied the directories
    masks, ids, images, img_id = [], [], [], 0

    def finish(group):
        """CAM + threshold for the queued batches (``streams`` of them in flight), then the host part per image."""
        if not group:
            return
        # device_batch = 0 (the default): one launch sequence per loader batch - the masks depend on the loader's batches
        # only, not on ``streams``, on how batches are dealt to ranks or on what else was queued (bit-reproducible).
        # device_batch = n > 0 (throughput option, 0.135 instead of 0.182 ms/img at n = 32): the loader's batches are merged
        # into device batches of n images.  A merged batch has its own per-tensor amax scales, tile and K-split choices, so
        # its masks equal the per-batch ones only outside the fp32 band of the network (~4e-3 of the CAM's range) and DO
        # depend on the flush threshold and the sharding - not for runs that must be reproducible
        if hasattr(layercam_gen, "generate_coalesced"):
            outs = layercam_gen.generate_coalesced([g[1] for g in group], alpha, [g[2] for g in group], cam_thresh, streams, device_batch)
        elif hasattr(layercam_gen, "generate_batches"):
            outs = layercam_gen.generate_batches([g[1] for g in group], alpha, [g[2] for g in group], cam_thresh, streams)
        else:
            outs = [layercam_gen.generate_batch(g[1], alpha=alpha, class_idx=g[2], thresh=cam_thresh) for g in group]
        for (imgs, _d, _l, first_id, take), (_cam, m) in zip(group, outs):
            if keep_largest_masks:
                m = keep_largest_batched(m)
            m_host = m if keep_on_device and not write_png else m.cpu().numpy()
            for i in range(take):
                gid = first_id + i
                mi = m_host[i]
                masks.append(mi)
                ids.append(gid)
                if keep_images:
                    images.append(imgs[i].detach())
                if write_png:
                    mt = torch.from_numpy(mi).float().unsqueeze(0).expand(3, -1, -1)
                    Image.fromarray(_to_png_u8(mt)).save(os.path.join(mask_dir, f"{gid}.png"))
                    im = imgs[i].detach().cpu().clone()
                    im = (im - im.min()) / (im.max() - im.min())
                    Image.fromarray(_to_png_u8(im)).save(os.path.join(image_dir, f"{gid}.png"))

    group = []
    for j, (imgs, (labels, _)) in enumerate(loader):
        if img_id >= max_images:
            break
        take = min(imgs.size(0), max_images - img_id)          # PsuedoMasks.py:49 - the 500-image cap
        if j % world != rank:
            img_id += take
            continue
        imgs_d = _to_device_async(imgs[:take], device)
        labels_d = _to_device_async(torch.as_tensor(labels[:take]), device)
        group.append((imgs, imgs_d, labels_d, img_id, take))
        img_id += take
        queued = sum(g[4] for g in group)
        if (device_batch > 0 and queued >= device_batch * max(1, streams)) or (device_batch <= 0 and len(group) >= max(1, streams)):
            finish(group)
            group = []
    finish(group)
    generate_pseudo_masks.last_masks = masks
    generate_pseudo_masks.last_ids = ids
    generate_pseudo_masks.last_images = images
    return image_dir, mask_dir


generate = generate_pseudo_masks   # north-star alias "PsuedoMasks.generate"

_MEAN = (0.485, 0.456, 0.406)
_STD = (0.229, 0.224, 0.225)


_norm_cache = {}


def _norm_constants(device):
    """ImageNet mean / std on ``device``, made once: ``torch.tensor(..., device=...)`` is a copy from pageable host memory
    and makes the host wait for the stream - per call it put stage 1 -> stage 2 in lockstep with the device."""
    key = str(torch.device(device))
    if key not in _norm_cache:
        _norm_cache[key] = (torch.tensor(_MEAN, device=device).view(1, 3, 1, 1), torch.tensor(_STD, device=device).view(1, 3, 1, 1))
    return _norm_cache[key]


def nearest_resize_index(n_out, n_in, device):
    """PIL NEAREST source index: floor((i + 0.5) * n_in / n_out)."""
    return ((torch.arange(n_out, device=device, dtype=torch.float64) + 0.5) * (n_in / n_out)).floor().long().clamp_(max=n_in - 1)


@torch.no_grad()
def stage_handoff(images, masks, size=(256, 256), device="cuda"):
    """In-memory stage-1 -> stage-2 hand-off: what ``PseudoSegmentationDataset`` (SegmentationDataset.py:19-38) would
    read back from the PNGs ``generate_pseudo_masks`` writes (PsuedoMasks.py:68-74), without the file system.

    images (N,3,h,w) float (stage-1 inputs), masks (N,h,w) uint8 {0,1} ->
      images256 (N,3,H,W) float32 on ``device``: min-max rescale per image, 8-bit quantisation (save_image),
                bilinear resize (PIL BILINEAR up-sampling == align_corners=False; HIP kernel), /255, ImageNet normalise;
      masks256  (N,H,W) uint8 {0,255}: x255 (save_image of a 0/1 tensor), NEAREST resize.
    """
    x = torch.as_tensor(images).to(device=device, dtype=torch.float32)
    lo = x.amin(dim=(1, 2, 3), keepdim=True)
    hi = x.amax(dim=(1, 2, 3), keepdim=True)
    q = ((x - lo) / (hi - lo)).mul(255).add(0.5).clamp(0, 255).floor()          # uint8 values, kept as float
    H, W = size
    # PIL resamples 8-bit images in two passes - columns first, then rows - and rounds to 8 bits after each
    if q.shape[-1] != W:
        q = ops.bilinear_resize(q.contiguous(), (q.shape[-2], W)).add(0.5).floor().clamp(0, 255)
    if q.shape[-2] != H:
        q = ops.bilinear_resize(q.contiguous(), (H, W)).add(0.5).floor().clamp(0, 255)
    mean, std = _norm_constants(device)
    img = (q / 255.0 - mean) / std
    if isinstance(masks, (list, tuple)) and len(masks) and torch.is_tensor(masks[0]):
        m = torch.stack(list(masks)).to(device)                        # device masks (keep_on_device): no host round trip
    else:
        m = torch.as_tensor(masks if torch.is_tensor(masks) else np.asarray(masks)).to(device)
    m = (m != 0).to(torch.uint8) * 255
    if tuple(m.shape[-2:]) != (H, W):
        ih = nearest_resize_index(H, m.shape[-2], device)
        iw = nearest_resize_index(W, m.shape[-1], device)
        m = m[:, ih][:, :, iw]
    return img.contiguous(), m.contiguous()
