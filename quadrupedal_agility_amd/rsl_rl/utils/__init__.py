from .utils import Normalizer, RunningMeanStd, TorchNormalizer, quaternion_slerp  # noqa: F401
