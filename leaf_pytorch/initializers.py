from leaf_pytorch_amd.initializers import GaborInit  # noqa: F401
