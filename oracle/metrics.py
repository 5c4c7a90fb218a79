"""Oracle metrics (TEST INFRASTRUCTURE ONLY).  Reference TraditionalModel/ExtraUtilities.py:4-21."""


def compute_iou_and_acc(pred_mask, true_mask):
    p, t = pred_mask > 0, true_mask > 0
    inter = int((p & t).sum())
    union = int((p | t).sum())
    acc = int((pred_mask == true_mask).sum()) / true_mask.numel()
    return inter / (union + 1e-8), acc
