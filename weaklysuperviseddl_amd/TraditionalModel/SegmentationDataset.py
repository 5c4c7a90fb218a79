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
