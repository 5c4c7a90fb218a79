"""Lovasz-softmax on the HIP path (reference TraditionalModel/LossFunctions/Lovasz-Softmax_Loss.py - the file name there
carries a hyphen and cannot be imported as a module; SegmentationModel.py:103-105 calls it as
``lovasz_softmax(F.softmax(outputs, dim=1), masks, classes='present', per_image=False, ignore=None)``).

``lovasz_softmax(probas, labels, classes='present', per_image=False, ignore=None)``: same arguments and meaning
(``classes`` 'present' or 'all'; an explicit class list is not supported).  The multi-class path only: the binary hinge
variants of the reference file (``lovasz_hinge``, ``binary_xloss``) have no caller.  Sorting, the Jaccard gradient and the
dot product run on the device (csrc/lovasz.hip); ties between equal errors are ranked by pixel index, the loss does not
depend on that order."""
from ... import ops


def lovasz_softmax(probas, labels, classes="present", per_image=False, ignore=None):
    return ops.lovasz_softmax(probas, labels, classes=classes, per_image=per_image, ignore=ignore)
