from leaf_pytorch_amd.frontend import Leaf, SquaredModulus  # noqa: F401
