from leaf_pytorch_amd.frontend_helper import get_frontend  # noqa: F401
