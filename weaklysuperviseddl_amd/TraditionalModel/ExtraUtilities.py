"""compute_iou_and_acc - reference TraditionalModel/ExtraUtilities.py:4-21 (host-side metric)."""


def compute_iou_and_acc(pred_mask, true_mask):
    pred_fg, true_fg = pred_mask > 0, true_mask > 0
    inter = (pred_fg & true_fg).sum().item()
    union = (pred_fg | true_fg).sum().item()
    correct = (pred_mask == true_mask).sum().item()
    return inter / (union + 1e-8), correct / true_mask.numel()
