"""feats2joints on the GPU: the step the reference runs on the CPU right after the hot path
(`joints = self.feats2joints(feats_rst.detach().cpu())`, ladiff.py:307; HumanML3D.py:44-48, Kit.py:48-53)."""
import torch

from . import _lib


class Feats2Joints:
    """`Feats2Joints(mean, std, njoints)(features[B,F,C]) -> joints[B,F,njoints,3]`, on the device the features are on.

    Drop-in for `datamodule.feats2joints`: build it from the datamodule's `hparams.mean / hparams.std / njoints`
    (`Feats2Joints.from_datamodule`).  No CPU implementation: CPU tensors raise."""

    def __init__(self, mean, std, njoints):
        self.mean = torch.as_tensor(mean, dtype=torch.float32).contiguous()
        self.std = torch.as_tensor(std, dtype=torch.float32).contiguous()
        self.njoints = int(njoints)
        if self.mean.shape != self.std.shape or self.mean.dim() != 1:
            raise ValueError("mean / std must be 1-D and of equal length")
        if self.mean.numel() < 4 + 3 * (self.njoints - 1):
            raise ValueError(f"{self.mean.numel()} features cannot hold {self.njoints} joints")
        self._dev = {}

    @classmethod
    def from_datamodule(cls, dm):
        hp = getattr(dm, "hparams", None)
        mean = getattr(hp, "mean", None) if hp is not None else None
        std = getattr(hp, "std", None) if hp is not None else None
        if mean is None:                       # Kit.py:52 keeps them as attributes
            mean, std = getattr(dm, "mean", None), getattr(dm, "std", None)
        if mean is None or std is None or not hasattr(dm, "njoints"):
            return None
        return cls(mean, std, dm.njoints)

    def __call__(self, features):
        if features.dim() != 3 or features.shape[-1] != self.mean.numel():
            raise ValueError(f"features {tuple(features.shape)} do not match {self.mean.numel()} feature statistics")
        dev = features.device
        if dev not in self._dev:
            self._dev[dev] = (self.mean.to(dev), self.std.to(dev))
        mean, std = self._dev[dev]
        x = features.detach().to(torch.float32).contiguous()
        B, F, C = x.shape
        joints = torch.empty(B, F, self.njoints, 3, dtype=torch.float32, device=dev)
        _lib.check(_lib.lib().ladiff_feats2joints(_lib.ptr(x), _lib.ptr(mean), _lib.ptr(std), B, F, C, self.njoints,
                                                  _lib.ptr(joints), _lib.stream_ptr()))
        return joints.to(features.dtype)
