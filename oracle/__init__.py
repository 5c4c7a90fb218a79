"""CPU oracle for the weakly-supervised segmentation hot path.

TEST INFRASTRUCTURE ONLY.  This package is a plain PyTorch-CPU restatement of
the arithmetic the reference (alexncoleman/WeaklySupervisedDL, directory
``TraditionalModel/``) executes on the path named by BASELINE.json.  It is the
checker for the HIP kernels, never the thing that is shipped or measured:

  * only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg
    of ``bench.py`` may import it;
  * nothing under ``weaklysuperviseddl_amd/`` imports it, and the product path
    raises when the HIP extension is absent instead of falling back to it.

Pinning status (see DESIGN.md "Oracle"):
  * losses / LayerCAM epilogues / keep_largest / refine_pseudo_mask /
    compute_iou_and_acc are pinned against golden vectors produced by
    executing the reference's own function bodies in the build container
    (``tests/golden/make_golden.py``; fixtures in ``tests/golden/*.npz``).
  * the two models (torchvision ResNet-50 / DeepLabV3-ResNet50) live in a
    third-party dependency (torchvision, version unpinned by the reference and
    absent from this image).  Their restatement follows the published
    architecture and is pinned only by structural self-checks (parameter
    counts 25,557,032 / 42,004,074, state_dict key sets): "parity unpinned"
    for the model weights/activations themselves.
"""
from .models import (  # noqa: F401
    ResNet50Trunk,
    FrozenResNetCAM,
    DeepLabV3ResNet50,
    build_segmentation_model,
)
from .losses import (  # noqa: F401
    LocalNormalizedCutLoss,
    ConstrainToBoundaryLossSingle,
    compute_affinities,
    pairwise_affinity_loss,
    lovasz_grad,
    lovasz_softmax,
    lovasz_softmax_flat,
)
from .layercam import LayerCAMGenerator, CAMGenerator, layercam_epilogue  # noqa: F401
from .pseudo_masks import keep_largest, cam_to_mask, generate_pseudo_masks  # noqa: F401
from .refine import refine_pseudo_mask  # noqa: F401
from .metrics import compute_iou_and_acc  # noqa: F401
