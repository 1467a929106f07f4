from leaf_pytorch_amd.modules import ExponentialMovingAverage, PCENLayer  # noqa: F401
