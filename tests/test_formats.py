"""the data formats either side of the hot path - pseudo-mask / image PNG round trip through
``PseudoSegmentationDataset`` and state_dict compatibility (torchvision key names, torch.save / weights_only
reload as the reference does at AlternatingDirectionCutLoss.py:483-492).  CPU only."""
import numpy as np
import torch
from PIL import Image


def test_png_round_trip(tmp_path):
    from weaklysuperviseddl_amd.TraditionalModel import PseudoSegmentationDataset
    from weaklysuperviseddl_amd.TraditionalModel.PsuedoMasks import _to_png_u8
    rng = np.random.RandomState(3)
    img_dir, mask_dir = tmp_path / "images_r", tmp_path / "pseudo_masks_r"
    img_dir.mkdir(), mask_dir.mkdir()
    masks, imgs = [], []
    for i in range(3):
        m = np.zeros((224, 224), np.uint8)
        m[30 + 10 * i:150, 40:120 + 20 * i] = 1
        masks.append(m)
        im = torch.from_numpy(rng.rand(3, 224, 224).astype(np.float32))
        imgs.append(im)
        mt = torch.from_numpy(m).float().unsqueeze(0).expand(3, -1, -1)           # save_image writes 3 equal channels
        Image.fromarray(_to_png_u8(mt)).save(mask_dir / f"{i}.png")
        Image.fromarray(_to_png_u8(im)).save(img_dir / f"{i}.png")
    ds = PseudoSegmentationDataset(str(img_dir), str(mask_dir), transform=True)
    assert len(ds) == 3
    for i in range(3):
        x, m = ds[i]
        assert x.shape == (3, 256, 256) and x.dtype == torch.float32
        assert m.shape == (256, 256) and m.dtype == torch.int64 and set(m.unique().tolist()) <= {0, 255}
        # NEAREST 224 -> 256 of the stored 0/255 mask; clamp(max=1) / (== 255) give the same binary mask
        want = np.array(Image.fromarray(masks[i] * 255).resize((256, 256), Image.NEAREST))
        assert np.array_equal(m.numpy(), want)
        assert torch.equal(torch.clamp(m, max=1), (m == 255).long())
        # image: quantised to 8 bit by the PNG, bilinear resize, ImageNet normalisation
        back = x * torch.tensor([0.229, 0.224, 0.225]).view(3, 1, 1) + torch.tensor([0.485, 0.456, 0.406]).view(3, 1, 1)
        assert 0.0 <= back.min() + 1e-6 and back.max() <= 1.0 + 1e-6
        assert abs(back.mean().item() - imgs[i].mean().item()) < 5e-3
    x, m, name = PseudoSegmentationDataset(str(img_dir), str(mask_dir), transform=True, return_name=True)[1]
    assert name == "1.png"
    raw_img, raw_mask = PseudoSegmentationDataset(str(img_dir), str(mask_dir))[0]
    assert raw_img.size == (224, 224) and raw_mask.mode == "L"


def test_state_dict_files_round_trip(tmp_path):
    """Checkpoints written by either implementation load into the other (same keys, shapes, dtypes)."""
    import oracle
    from weaklysuperviseddl_amd.TraditionalModel import FrozenResNetCAM, build_segmentation_model
    torch.manual_seed(0)
    cls_ref = oracle.FrozenResNetCAM(37)
    path = tmp_path / "classifier_weights_without_affinity.pth"
    torch.save(cls_ref.state_dict(), path)
    mine = FrozenResNetCAM(37)
    missing, unexpected = mine.load_state_dict(torch.load(path, weights_only=True))
    assert not missing and not unexpected
    for k, v in cls_ref.state_dict().items():
        assert torch.equal(mine.state_dict()[k], v), k
    seg = build_segmentation_model()
    path2 = tmp_path / "deeplab.pth"
    torch.save(seg.state_dict(), path2)
    ref = oracle.build_segmentation_model()
    ref.load_state_dict(torch.load(path2, weights_only=True))
    keys = list(seg.state_dict())
    assert keys == list(ref.state_dict())                 # same ORDER too (torchvision registration order)
    assert keys[0] == "backbone.conv1.weight" and "classifier.0.convs.4.1.weight" in keys and "aux_classifier.4.bias" in keys


def test_product_compute_iou_and_acc_matches_the_reference_vectors(golden):
    """The product's metric (TraditionalModel/ExtraUtilities.py) against the golden vector of the reference body
    (ExtraUtilities.py:4-21) - host code, no kernel involved."""
    import torch
    from weaklysuperviseddl_amd.TraditionalModel import compute_iou_and_acc
    g = golden("refine_metrics")
    iou, acc = compute_iou_and_acc(torch.from_numpy(g["metric_pred"]), torch.from_numpy(g["metric_true"]))
    assert abs(iou - g["metric_iou_acc"][0]) < 1e-12 and abs(acc - g["metric_iou_acc"][1]) < 1e-12
    z = torch.zeros(4, 4, dtype=torch.long)
    assert compute_iou_and_acc(z, z) == (0.0, 1.0)                 # empty union: 0 / (0 + 1e-8)
