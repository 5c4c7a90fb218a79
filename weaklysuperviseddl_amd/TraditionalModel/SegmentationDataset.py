"""PseudoSegmentationDataset - the on-disk hand-off between stage 1 and stage 2.

Mirrors reference TraditionalModel/SegmentationDataset.py:8-39 (and the notebook twin
AlternatingDirectionCutLoss.py:431-466, which also returns the file name): sorted ``listdir`` pairing of
``images/`` and ``pseudo_masks/``; ``joint_transform`` = resize to 256x256 (bilinear image, NEAREST mask),
to-tensor, ImageNet normalise, mask -> int64.  Host-side IO only (PIL); torchvision is not needed.
Mask PNGs hold {0, 255} (``save_image`` of a 0/1 tensor); training clamps them to {0, 1}
(AlternatingDirectionCutLoss.py:695), refinement decodes ``mask == 255`` (:726).
"""
import os

import numpy as np
import torch
from PIL import Image
from torch.utils.data import Dataset

_MEAN = torch.tensor([0.485, 0.456, 0.406]).view(3, 1, 1)
_STD = torch.tensor([0.229, 0.224, 0.225]).view(3, 1, 1)


class PseudoSegmentationDataset(Dataset):
    def __init__(self, img_dir, mask_dir, transform=False, return_name=False, size=(256, 256)):
        self.img_dir, self.mask_dir = img_dir, mask_dir
        self.image_list = sorted(os.listdir(img_dir))
        self.mask_list = sorted(os.listdir(mask_dir))
        self.transform, self.return_name, self.size = transform, return_name, tuple(size)

    def __len__(self):
        return len(self.image_list)

    def joint_transform(self, image, mask):
        w_h = (self.size[1], self.size[0])
        image = image.resize(w_h, Image.BILINEAR)
        mask = mask.resize(w_h, Image.NEAREST)
        img = torch.from_numpy(np.asarray(image, dtype=np.uint8).copy()).permute(2, 0, 1).float().div(255.0)
        img = (img - _MEAN) / _STD
        return img, torch.as_tensor(np.array(mask), dtype=torch.long)

    def __getitem__(self, idx):
        image = Image.open(os.path.join(self.img_dir, self.image_list[idx])).convert("RGB")
        mask = Image.open(os.path.join(self.mask_dir, self.mask_list[idx])).convert("L")
        if self.transform:
            image, mask = self.joint_transform(image, mask)
        return (image, mask, self.mask_list[idx]) if self.return_name else (image, mask)


class InMemoryPseudoDataset(Dataset):
    """The same items ``PseudoSegmentationDataset(..., transform=True, return_name=True)`` yields, held as device
    tensors: ``images`` (N,3,H,W) float32 normalised, ``masks`` (N,H,W) uint8 {0,255}, ``names`` (e.g. "17.png").
    Built by ``PsuedoMasks.stage_handoff`` from stage 1's in-memory output; ``set_masks`` is the in-memory twin of
    overwriting the mask PNGs (reference AlternatingDirectionCutLoss.py:806-809).  ``batches`` iterates like
    ``DataLoader(dataset, batch_size, shuffle)`` without leaving the device."""

    def __init__(self, images, masks, names=None):
        assert images.shape[0] == masks.shape[0] and images.shape[-2:] == masks.shape[-2:]
        self.images, self.masks = images, masks.to(torch.uint8)
        self.names = list(names) if names is not None else [f"{i}.png" for i in range(images.shape[0])]

    def __len__(self):
        return self.images.shape[0]

    def __getitem__(self, idx):
        return self.images[idx], self.masks[idx].long(), self.names[idx]

    def set_masks(self, idx, refined):
        """refined: (n,H,W) float/bool masks in {0,1} (refine_pseudo_mask's output) -> stored as {0,255}, what
        ``save_image`` + ``Image.open(...).convert('L')`` turn them into."""
        self.masks[idx] = (refined > 0).to(torch.uint8) * 255

    def num_batches(self, batch_size, drop_single=True):
        n, r = divmod(len(self), batch_size)
        return n + (1 if r > (1 if drop_single else 0) else 0)

    def batches(self, batch_size, shuffle=True, generator=None, limit=None):
        """Yields (images, masks int64, index tensor).  A trailing batch of ONE image is skipped like the reference's
        trainer does (SegmentationModel.py:97-98: train-mode BN cannot normalise one pooled value)."""
        n = len(self)
        order = torch.randperm(n, generator=generator) if shuffle else torch.arange(n)
        order = order.to(self.images.device)
        done = 0
        for s in range(0, n, batch_size):
            idx = order[s:s + batch_size]
            if idx.numel() == 1 and batch_size > 1:
                continue
            if limit is not None and done >= limit:
                return
            done += 1
            yield self.images[idx], self.masks[idx].long(), idx
