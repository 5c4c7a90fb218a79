"""Oracle pseudo-mask generation (TEST INFRASTRUCTURE ONLY - see oracle/__init__.py).

Restates reference TraditionalModel/PsuedoMasks.py:
  * ``keep_largest``            :15-21  (skimage ``label`` default = 8-connectivity, raster label
    order; ``max(regions, key=area)`` keeps the FIRST label among equal areas; empty -> input)
  * ``generate_pseudo_masks``   :23-79  (per image: CAM -> zero below thresh -> ``> 0`` -> uint8
    -> optional keep_largest -> mask PNG 0/255 (3 identical channels, as torchvision
    ``save_image`` writes a 1-channel tensor) + min-max rescaled image PNG; 500-image cap :49)
skimage is absent from this image; scipy.ndimage.label with a 3x3 structure labels 8-connected
components in the same raster order (pinned against skimage 0.18.3 outputs in tests/golden).
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


def cam_to_mask(cam, cam_thresh=0.3):
    """PsuedoMasks.py:59-62: ``cam[cam < t] = 0; (cam > 0)`` -> uint8 (H,W) ndarray."""
    c = cam.detach().clone()
    c[c < cam_thresh] = 0.0
    return (c.cpu().numpy() > 0).astype(np.uint8)


def _to_png_u8(t):
    # torchvision.utils.save_image: mul(255).add(0.5).clamp(0,255) -> uint8, HWC
    return t.mul(255).add(0.5).clamp(0, 255).permute(1, 2, 0).to(torch.uint8).cpu().numpy()


def generate_pseudo_masks(loader, layercam_gen, cam_thresh=0.3, alpha=1.0, keep_largest_masks=True,
                          run_id="default", out_root="/content", max_images=500, write_png=True):
    """Returns (image_dir, mask_dir); also stores the in-memory uint8 masks in ``.last_masks``."""
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
        for i in range(imgs.size(0)):
            if img_id >= max_images:
                break
            cam = layercam_gen.generate(imgs[i], alpha=alpha,
                                        class_idx=torch.tensor([int(labels[i])]))
            m = cam_to_mask(cam.squeeze(0), cam_thresh)
            if keep_largest_masks:
                m = keep_largest(m)
            masks.append(m)
            if write_png:
                mt = torch.from_numpy(m).float().unsqueeze(0).expand(3, -1, -1)
                Image.fromarray(_to_png_u8(mt)).save(os.path.join(mask_dir, f"{img_id}.png"))
                im = imgs[i].detach().cpu().clone()
                im = (im - im.min()) / (im.max() - im.min())
                Image.fromarray(_to_png_u8(im)).save(os.path.join(image_dir, f"{img_id}.png"))
            img_id += 1
    generate_pseudo_masks.last_masks = masks
    return image_dir, mask_dir
