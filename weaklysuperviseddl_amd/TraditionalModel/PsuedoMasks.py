"""Pseudo-mask generation on the HIP path.

Mirrors reference TraditionalModel/PsuedoMasks.py: ``keep_largest`` (:15-21) and
``generate_pseudo_masks`` (:23-79).  Differences, all outside the arithmetic:
  * CAM + threshold run batched on the device (``LayerCAMGenerator.generate_batch`` with the
    threshold fused into the epilogue kernel) - one device->host copy of uint8 masks per batch instead
    of a float map per image;
  * output directories are parameters (the reference hard-codes /content/...); ``write_png=False``
    keeps the masks in memory (``generate_pseudo_masks.last_masks``) for the in-memory hand-off to
    stage 2;
  * ``keep_largest`` stays on the host as in the reference (skimage there; scipy.ndimage here -
    8-connectivity, raster label order, first label wins area ties, empty mask returned unchanged).
"""
import os

import numpy as np
import torch
from scipy import ndimage

_EIGHT = np.ones((3, 3), dtype=bool)


def keep_largest(mask):
    lab, n = ndimage.label(np.asarray(mask) != 0, structure=_EIGHT)
    if n == 0:
        return mask
    areas = np.bincount(lab.ravel(), minlength=n + 1)[1:]
    return (lab == (int(np.argmax(areas)) + 1)).astype(np.uint8)


def _to_png_u8(t):
    return t.mul(255).add(0.5).clamp(0, 255).permute(1, 2, 0).to(torch.uint8).cpu().numpy()


def generate_pseudo_masks(loader, layercam_gen, cam_thresh=0.3, alpha=1.0, keep_largest_masks=True,
                          run_id="default", out_root="/content", max_images=500, write_png=True,
                          device="cuda"):
    mask_dir = os.path.join(out_root, f"pseudo_masks_{run_id}")
    image_dir = os.path.join(out_root, f"images_{run_id}")
    if write_png:
        from PIL import Image
        for d in (mask_dir, image_dir):
            os.makedirs(d, exist_ok=True)
            for f in os.listdir(d):
                os.remove(os.path.join(d, f))
    masks, img_id = [], 0
    for imgs, (labels, _) in loader:
        if img_id >= max_images:
            break
        take = min(imgs.size(0), max_images - img_id)
        imgs_d = imgs[:take].to(device, non_blocking=True)
        labels_d = torch.as_tensor(labels[:take]).to(device)
        _cam, m = layercam_gen.generate_batch(imgs_d, alpha=alpha, class_idx=labels_d, thresh=cam_thresh)
        m_host = m.cpu().numpy()
        for i in range(take):
            mi = keep_largest(m_host[i]) if keep_largest_masks else m_host[i]
            masks.append(mi)
            if write_png:
                mt = torch.from_numpy(mi).float().unsqueeze(0).expand(3, -1, -1)
                Image.fromarray(_to_png_u8(mt)).save(os.path.join(mask_dir, f"{img_id}.png"))
                im = imgs[i].detach().cpu().clone()
                im = (im - im.min()) / (im.max() - im.min())
                Image.fromarray(_to_png_u8(im)).save(os.path.join(image_dir, f"{img_id}.png"))
            img_id += 1
    generate_pseudo_masks.last_masks = masks
    return image_dir, mask_dir


generate = generate_pseudo_masks   # north-star alias "PsuedoMasks.generate"
