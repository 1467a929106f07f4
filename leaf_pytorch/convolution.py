from leaf_pytorch_amd.modules import GaborConstraint, GaborConv1d  # noqa: F401
