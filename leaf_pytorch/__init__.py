"""Import-path shim: lets code written against the reference (``from leaf_pytorch import get_frontend``,
``from leaf_pytorch.frontend import Leaf`` -- e.g. models/classifier.py:3, test_leaf.py:2) pick up the
MI355X-native implementation in ``leaf_pytorch_amd`` without edits."""
from leaf_pytorch_amd.frontend_helper import get_frontend  # noqa: F401
