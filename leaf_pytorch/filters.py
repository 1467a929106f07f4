from leaf_pytorch_amd.initializers import GaborFilter  # noqa: F401
