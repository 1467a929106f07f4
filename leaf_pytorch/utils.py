from leaf_pytorch_amd.modules import get_padding_value  # noqa: F401
