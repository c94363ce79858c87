"""Shared bases: parameter-holding nn.Modules whose arithmetic runs in libcmk_hip.so.

Modules keep their parameters under the reference's state-dict names/shapes (so reference checkpoints load) and
derive device-resident packed weights lazily; any `load_state_dict` / `.to()` / `_apply` drops the packed cache.
"""
from typing import Dict

import torch
from torch import nn

from ..structures import ShapeSpec


class HipModule(nn.Module):
    def __init__(self):
        super().__init__()
        self._packed_cache = None

    def invalidate_packed(self) -> None:
        for m in self.modules():
            if isinstance(m, HipModule):
                m._packed_cache = None

    def _apply(self, fn, *a, **k):
        self.invalidate_packed()
        return super()._apply(fn, *a, **k)

    def _load_from_state_dict(self, *a, **k):
        self._packed_cache = None
        return super()._load_from_state_dict(*a, **k)

    def load_state_dict(self, *a, **k):
        self.invalidate_packed()
        return super().load_state_dict(*a, **k)

    def packed(self):
        if self._packed_cache is None:
            dev = next(self.parameters()).device if any(True for _ in self.parameters()) else next(self.buffers()).device
            if dev.type != "cuda":
                from .._lib import CmkError
                raise CmkError("{} is on {}: the CenterMask2 path runs as HIP kernels on an MI355X; there is no CPU "
                               "fallback (move the model with .to('cuda'))".format(type(self).__name__, dev))
            with torch.no_grad():
                self._packed_cache = self._build_packed(dev)
        return self._packed_cache

    def _build_packed(self, device):
        raise NotImplementedError


class Backbone(HipModule):
    """detectron2 `Backbone` surface: forward -> dict of features, output_shape(), size_divisibility."""

    def output_shape(self) -> Dict[str, ShapeSpec]:
        return {name: ShapeSpec(channels=self._out_feature_channels[name], stride=self._out_feature_strides[name])
                for name in self._out_features}

    @property
    def size_divisibility(self) -> int:
        return 0


class FrozenBatchNorm2d(nn.Module):
    """Parameter holder with detectron2's FrozenBatchNorm2d buffers (eps 1e-5); folded into the conv epilogue."""

    def __init__(self, num_features: int, eps: float = 1e-5):
        super().__init__()
        self.num_features = num_features
        self.eps = eps
        self.register_buffer("weight", torch.ones(num_features))
        self.register_buffer("bias", torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features) - eps)


def get_norm(norm: str, out_channels: int):
    """detectron2's get_norm for the names an inference path can meet: "" -> None, "FrozenBN", "BN" / "SyncBN" (their eval form: running
    statistics), "GN" (32 groups).  The returned module only HOLDS the parameters (state-dict names of d2's Conv2d wrapper: `<conv>.norm.*`);
    the arithmetic is folded into the conv's epilogue (batch norms) or runs as the GroupNorm kernel."""
    if not norm:
        return None
    if norm == "FrozenBN":
        return FrozenBatchNorm2d(out_channels)
    if norm in ("BN", "SyncBN"):
        return nn.BatchNorm2d(out_channels)
    if norm == "GN":
        return nn.GroupNorm(32, out_channels)
    raise NotImplementedError("norm '{}'".format(norm))


class NormConv2d(nn.Conv2d):
    """nn.Conv2d with detectron2's optional `norm` child (d2 `Conv2d(..., norm=get_norm(name, C))`: parameters `weight`[, `bias`], `norm.*`)."""

    def __init__(self, *args, norm=None, **kwargs):
        super().__init__(*args, **kwargs)
        self.norm = norm


def fold_norm(conv: nn.Conv2d):
    """(scale, shift, gn) of a conv's epilogue: batch norms fold with the bias into per-channel scale/shift (eval statistics);
    a GroupNorm stays a pass of its own: gn = (gamma, beta, eps, groups), applied to conv(x) + bias."""
    norm = getattr(conv, "norm", None)
    bias = conv.bias.detach().float().cpu() if conv.bias is not None else None
    if norm is None:
        return None, bias, None
    if isinstance(norm, nn.GroupNorm):
        return None, bias, (norm.weight.detach().float(), norm.bias.detach().float(), norm.eps, norm.num_groups)
    w, b = norm.weight.detach().double().cpu(), norm.bias.detach().double().cpu()
    scale = w / torch.sqrt(norm.running_var.detach().double().cpu() + norm.eps)
    shift = b - norm.running_mean.detach().double().cpu() * scale
    if bias is not None:
        shift = shift + bias.double() * scale
    return scale.float(), shift.float(), None
