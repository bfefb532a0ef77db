from .iid_dataset import IIDSampler

__all__ = ["IIDSampler"]
