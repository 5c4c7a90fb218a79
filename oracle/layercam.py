"""Oracle LayerCAM (TEST INFRASTRUCTURE ONLY - see oracle/__init__.py).

Restates ``LayerCAMGenerator``:
  * variant "modular"  - reference TraditionalModel/LayerCAM.py:7-81
  * variant "notebook" - reference TraditionalModel/AlternatingDirectionCutLoss.py:216-293
    (extra ``**alpha`` + second min-max per layer, no final power).

``layercam_epilogue`` is the pure function the HIP kernel ``wsdl_layercam_epilogue`` implements:
it takes the hooked activations / gradients and produces the (B, outH, outW) map.
"""
import torch
import torch.nn.functional as F


def _minmax_(c):
    c = c - c.amin(dim=(-2, -1), keepdim=True)
    return c / (c.amax(dim=(-2, -1), keepdim=True) + 1e-8)


def layercam_epilogue(acts, grads, out_hw=(224, 224), alpha=1.0, variant="modular"):
    """acts/grads: lists of (B,C_l,h_l,w_l) -> (B,outH,outW).

    modular  (LayerCAM.py:52-76): relu(sum_c relu(g*a)) -> min-max -> bilinear -> layer mean
                                   -> clamp(0)**alpha
    notebook (AlternatingDirectionCutLoss.py:261-284): ... -> min-max -> **alpha -> min-max
                                   -> bilinear -> layer mean
    """
    maps = []
    for a, g in zip(acts, grads):
        cam = F.relu(F.relu(g * a).sum(dim=1))
        cam = _minmax_(cam)
        if variant == "notebook":
            cam = _minmax_(cam ** alpha)
        elif variant != "modular":
            raise ValueError(variant)
        maps.append(F.interpolate(cam.unsqueeze(1), size=tuple(out_hw), mode="bilinear",
                                  align_corners=False).squeeze(1))
    out = maps[0]
    for m in maps[1:]:
        out = out + m
    out = out / len(maps)
    if variant == "modular":
        out = out.clamp(min=0.0) ** alpha
    return out


class LayerCAMGenerator:
    """Hook-based generator over any module exposing the named layers as attributes."""

    def __init__(self, model, target_layer_names=("layer3", "layer4"), variant="modular",
                 out_hw=(224, 224)):
        self.model = model.eval()
        self.target_layer_names = list(target_layer_names)
        self.variant = variant
        self.out_hw = tuple(out_hw)
        self.activations, self.gradients = {}, {}
        for name in self.target_layer_names:
            layer = getattr(self.model, name)
            layer.register_forward_hook(self._fwd(name))
            layer.register_full_backward_hook(self._bwd(name))

    def _fwd(self, name):
        def hook(_m, _inp, out):
            self.activations[name] = out
        return hook

    def _bwd(self, name):
        def hook(_m, _gin, gout):
            self.gradients[name] = gout[0]
        return hook

    def generate(self, images, alpha=1.0, class_idx=None):
        """images (3,H,W) [or (B,3,H,W)] -> (B,outH,outW).  Accepts both reference keyword orders."""
        if torch.is_tensor(alpha) and not torch.is_tensor(class_idx):
            # notebook order generate(images, class_idx, alpha)
            alpha, class_idx = (1.0 if class_idx is None else class_idx), alpha
        self.activations.clear()
        self.gradients.clear()
        x = images.unsqueeze(0) if images.dim() == 3 else images
        x = x.detach().clone().requires_grad_()
        with torch.enable_grad():
            logits, _ = self.model(x)
            if class_idx is None:
                class_idx = logits.argmax(dim=1)
            score = logits.gather(1, class_idx.view(-1, 1)).squeeze()
            score.backward(torch.ones_like(score), retain_graph=False)
        with torch.no_grad():
            acts = [self.activations[n].detach() for n in self.target_layer_names]
            grads = [self.gradients[n].detach() for n in self.target_layer_names]
            return layercam_epilogue(acts, grads, self.out_hw, alpha, self.variant)

    __call__ = generate

    def generate_bg_cam(self, image_tensor, valid_class_indices, alpha=2.0):
        """Notebook class only - reference AlternatingDirectionCutLoss.py:296-318: the LayerCAM of ``valid_class_indices``
        (``generate(image, valid_class_indices)`` = notebook order, alpha 1.0), maximum over its leading axis,
        ``1 - clamp(1 - max, 0) ** alpha`` as the background map, both bilinearly resized to 224 x 224."""
        all_cams = self.generate(image_tensor, 1.0, class_idx=torch.as_tensor(valid_class_indices))
        max_obj = all_cams.max(dim=0).values
        m_bg = 1.0 - ((1.0 - max_obj).clamp(min=0.0) ** alpha)
        up = lambda t: F.interpolate(t[None, None], size=(224, 224), mode="bilinear", align_corners=False).squeeze()
        return up(m_bg), up(max_obj)


class CAMGenerator:
    """Classic fc-weight CAM for every class - reference AlternatingDirectionCutLoss.py:320-403
    (``einsum("c,chw->hw")`` per class, ReLU, per-class min-max; ``generate_bg_cam``: masked max over the valid
    classes, ``1 - (1 - max)**alpha`` background map, both bilinearly resized to 224x224)."""

    def __init__(self, model):
        self.model = model.eval()

    def generate_all_cams(self, image_tensor):
        with torch.no_grad():
            _logits, feats = self.model(image_tensor.unsqueeze(0))
            f = feats[-1][0]                                                   # (Cf,h,w)
            cams = F.relu(torch.einsum("kc,chw->khw", self.model.fc.weight, f))
            return _minmax_(cams)

    def generate_bg_cam(self, image_tensor, valid_class_indices, alpha=1.0, out_hw=(224, 224)):
        cams = self.generate_all_cams(image_tensor)
        keep = torch.zeros(cams.shape[0], 1, 1, dtype=cams.dtype)
        keep[list(valid_class_indices)] = 1.0
        max_obj = (cams * keep).max(dim=0).values
        m_bg = 1.0 - ((1.0 - max_obj).clamp(min=0.0) ** alpha)
        up = lambda t: F.interpolate(t[None, None], size=tuple(out_hw), mode="bilinear", align_corners=False)[0, 0]
        return up(m_bg), up(max_obj)
