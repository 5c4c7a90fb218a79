"""ConstrainToBoundaryLossSingle on the HIP path.

Mirrors reference TraditionalModel/AlternatingDirectionBoundaryLoss.py:12-70 with the intended static
semantics of ``compute_affinities_single`` (the reference raises TypeError as written):
per image, input already a probability map (no softmax), colour + spatial Gaussian affinity, reflect
padding, ``sum_k mean_{H,W}(w_k * sum_c (p - p')^2) / K``.  One fused HIP kernel computes the loss and
d loss / d preds (``wsdl_pairwise_affinity_loss_fwd_bwd`` with apply_softmax=0, normalise=1).
"""
import torch.nn as nn

from .. import ops
from .AlternatingDirectionCutLoss import run_alternating_training  # noqa: F401  (reference :153-206 is the broken
#   modular re-write of the same loop - SURVEY.md D6; the working behaviour is the script's, implemented there)


class ConstrainToBoundaryLossSingle(nn.Module):
    def __init__(self, sigma_color=0.1, sigma_space=5, window_size=5, eps=1e-8):
        super().__init__()
        self.sigma_color, self.sigma_space = sigma_color, sigma_space
        self.window_size, self.eps = window_size, eps

    def forward(self, preds, image):
        """preds (C,H,W) probabilities, image (3,H,W) -> scalar.  Also accepts batches (B,C,H,W) -> (B,)."""
        single = preds.dim() == 3
        if single:
            preds, image = preds.unsqueeze(0), image.unsqueeze(0)
        out = ops.pairwise_affinity_loss(preds, image, self.window_size, self.sigma_color, self.sigma_space,
                                         apply_softmax=False, normalise=1)
        return out[0] if single else out

    @staticmethod
    def compute_affinities_single(image, sigma_color=0.1, sigma_space=5, window_size=5):
        a = ops.compute_affinities(image.unsqueeze(0), sigma_color, sigma_space, window_size)   # (K,1,1,H,W)
        return [a[k, 0] for k in range(a.shape[0])]                                            # K x (1,H,W)
