"""weaklysuperviseddl_amd - MI355X-native hot path of the weakly-supervised segmentation pipeline.

``TraditionalModel`` holds the drop-in surfaces (same names / signatures as the reference's
``TraditionalModel/*.py``); ``ops`` / ``nn`` are the tensor plumbing over libwsdl_hip.so (C ABI in
``include/wsdl_hip.h``).  There is no CPU fallback: without the built library, or with host tensors,
calls raise.
"""
from ._lib import LIB_PATH, WsdlError, lib  # noqa: F401

__version__ = "0.1.0"
