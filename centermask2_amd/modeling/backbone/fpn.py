"""FPN (detectron2 surface; constructed at vovnet.py:547-554) and the FCOS top blocks (backbone/fpn.py:17-53).

Kernels: lateral 1x1 convs run on the MFMA conv with the nearest-x2 top-down add fused into the epilogue
(res_mode 2), output 3x3 convs and the stride-2 P6/P7 convs on the same kernel; `relu(p6)` (fpn.py:34) is applied
while P7's input tile is staged.  Parameter names follow detectron2: fpn_lateral{3,4,5}, fpn_output{3,4,5},
top_block.p6/p7.
"""
import math

import torch
from torch import nn

from ... import ops
from ...ops import View
from ..base import Backbone, HipModule, NormConv2d, fold_norm, get_norm

__all__ = ["FPN", "LastLevelP6P7", "LastLevelP6", "LastLevelMaxPool"]


class LastLevelP6P7(nn.Module):
    def __init__(self, in_channels, out_channels, in_features="res5"):
        super().__init__()
        self.num_levels = 2
        self.in_feature = in_features
        self.p6 = nn.Conv2d(in_channels, out_channels, 3, 2, 1)
        self.p7 = nn.Conv2d(out_channels, out_channels, 3, 2, 1)


class LastLevelP6(nn.Module):
    def __init__(self, in_channels, out_channels, in_features="res5"):
        super().__init__()
        self.num_levels = 1
        self.in_feature = in_features
        self.p6 = nn.Conv2d(in_channels, out_channels, 3, 2, 1)


class LastLevelMaxPool(nn.Module):
    """d2's top block of the plain FPN backbone (vovnet.py:504-524): p6 = max_pool2d(p5, kernel 1, stride 2) — every second pixel of p5;
    no parameters."""

    def __init__(self):
        super().__init__()
        self.num_levels = 1
        self.in_feature = "p5"


class FPN(Backbone):
    def __init__(self, bottom_up, in_features, out_channels, norm="", top_block=None, fuse_type="sum"):
        super().__init__()
        if fuse_type != "sum":
            raise NotImplementedError("FPN fuse_type '{}' (the reference config uses 'sum')".format(fuse_type))
        input_shapes = bottom_up.output_shape()
        strides = [input_shapes[f].stride for f in in_features]
        for a, b in zip(strides[:-1], strides[1:]):
            assert b == 2 * a, "FPN input strides must double: {}".format(strides)
        self._stages = [int(math.log2(s)) for s in strides]
        for f, st in zip(in_features, self._stages):
            # d2 FPN: bias only without a norm; MODEL.FPN.NORM (vovnet.py:550) "" in the reference recipe, "GN" / "FrozenBN" / "BN" as get_norm
            self.add_module("fpn_lateral{}".format(st), NormConv2d(input_shapes[f].channels, out_channels, kernel_size=1, bias=norm == "",
                                                                    norm=get_norm(norm, out_channels)))
            self.add_module("fpn_output{}".format(st), NormConv2d(out_channels, out_channels, kernel_size=3, padding=1, bias=norm == "",
                                                                   norm=get_norm(norm, out_channels)))
        self.top_block = top_block
        self.in_features = tuple(in_features)
        self.bottom_up = bottom_up
        self._out_feature_strides = {"p{}".format(st): s for st, s in zip(self._stages, strides)}
        if top_block is not None:
            for s in range(self._stages[-1], self._stages[-1] + top_block.num_levels):
                self._out_feature_strides["p{}".format(s + 1)] = 2 ** (s + 1)
        self._out_features = list(self._out_feature_strides.keys())
        self._out_feature_channels = {k: out_channels for k in self._out_features}
        self._size_divisibility = strides[-1]
        self._fuse_type = fuse_type

    @property
    def size_divisibility(self):
        return self._size_divisibility

    def _build_packed(self, dev):
        P = {}
        for st in self._stages:
            lat, out = getattr(self, "fpn_lateral{}".format(st)), getattr(self, "fpn_output{}".format(st))
            for name, conv in (("lat", lat), ("out", out)):
                scale, shift, gn = fold_norm(conv)
                P["{}{}".format(name, st)] = ops.PackedConv(conv.weight, scale, shift, dev)
                P["{}{}_gn".format(name, st)] = None if gn is None else (gn[0].contiguous().to(dev), gn[1].contiguous().to(dev), gn[2], gn[3])
        if self.top_block is not None and not isinstance(self.top_block, LastLevelMaxPool):
            P["p6"] = ops.PackedConv(self.top_block.p6.weight, None, self.top_block.p6.bias, dev, stride=2)
            if self.top_block.num_levels == 2:
                P["p7"] = ops.PackedConv(self.top_block.p7.weight, None, self.top_block.p7.bias, dev, stride=2)
        return P

    def forward_views(self, x):
        P = self.packed()
        bu = self.bottom_up.forward_views(x)
        results = {}
        prev = None
        for f, st in zip(reversed(self.in_features), reversed(self._stages)):
            c = bu[f]
            if prev is not None and (prev.t.shape[1] * 2 != c.t.shape[1] or prev.t.shape[2] * 2 != c.t.shape[2]):
                raise ValueError("FPN: feature {} is {}x{}, not twice the level above ({}x{}); pad the input to a multiple of {}"
                                 .format(f, c.t.shape[1], c.t.shape[2], prev.t.shape[1], prev.t.shape[2], self._size_divisibility))
            lat_gn, out_gn = P["lat{}_gn".format(st)], P["out{}_gn".format(st)]
            if lat_gn is None:
                prev = ops.conv_out(c, P["lat{}".format(st)], res=prev, res_upsample=prev is not None)  # lateral + up2(top-down) in the epilogue
            else:                                           # a GroupNorm sits between the lateral conv and the sum
                lat = ops.conv_out(c, P["lat{}".format(st)])
                ops.groupnorm_relu_(lat.t, lat_gn[0], lat_gn[1], lat_gn[3], lat_gn[2], relu=False)
                if prev is not None:
                    ops.upsample2x_add_(lat, prev)
                prev = lat
            results["p{}".format(st)] = ops.conv_out(prev, P["out{}".format(st)])
            if out_gn is not None:
                ops.groupnorm_relu_(results["p{}".format(st)].t, out_gn[0], out_gn[1], out_gn[3], out_gn[2], relu=False)
        if self.top_block is not None:
            src_name = self.top_block.in_feature
            src = bu[src_name] if src_name in bu else results[src_name]
            top = self._stages[-1]
            p6 = ops.maxpool1x1s2(src) if isinstance(self.top_block, LastLevelMaxPool) else ops.conv_out(src, P["p6"])
            results["p{}".format(top + 1)] = p6
            if self.top_block.num_levels == 2:
                results["p{}".format(top + 2)] = ops.conv_out(p6, P["p7"], in_relu=True)     # p7(relu(p6)), fpn.py:34
        return {k: results[k] for k in self._out_features}

    def forward(self, x):
        return {k: v.nchw() for k, v in self.forward_views(x).items()}
