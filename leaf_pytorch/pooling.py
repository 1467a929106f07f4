from leaf_pytorch_amd.modules import GaussianLowPass  # noqa: F401
