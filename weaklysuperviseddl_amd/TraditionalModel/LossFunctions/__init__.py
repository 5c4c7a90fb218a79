from .Lovasz_Softmax_Loss import lovasz_softmax  # noqa: F401
