from .dataset import BaseDataSets, DeviceLoader, RandomGenerator, TwoStreamBatchSampler  # noqa: F401
