"""Drop-in surfaces mirroring the reference's ``TraditionalModel/`` modules (same public names)."""
from .ClassificationModel import FrozenResNetCAM, train_fc_only, evaluate_classification  # noqa: F401
from .LayerCAM import LayerCAMGenerator, CAMGenerator, evaluate_layercam_on_test_set  # noqa: F401
from .PsuedoMasks import generate_pseudo_masks, keep_largest, generate, stage_handoff  # noqa: F401
from .SegmentationModel import (SegmentationModel, build_segmentation_model, train_step, evaluate_model,  # noqa: F401
                                train_segmentation_model)
from .AlternatingDirectionCutLoss import (  # noqa: F401
    LocalNormalizedCutLoss, compute_affinities, refine_pseudo_mask, refine_pseudo_masks_batched, train_model,
    refine_dataset, run_alternating_training, network_soft_prediction)
from .AlternatingDirectionBoundaryLoss import ConstrainToBoundaryLossSingle  # noqa: F401
from .ExtraUtilities import compute_iou_and_acc  # noqa: F401
from .SegmentationDataset import PseudoSegmentationDataset, InMemoryPseudoDataset  # noqa: F401
