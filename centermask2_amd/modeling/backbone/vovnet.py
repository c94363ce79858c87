"""VoVNetV2-eSE backbone on HIP kernels, behind the reference's builder names.

Mirrors centermask2/centermask/modeling/backbone/vovnet.py: stage specs :60-108, conv3x3/conv1x1 (conv, FrozenBN,
ReLU) :205-236, eSEModule :247-260, _OSA_module :263-332, _OSA_stage :335-376, VoVNet :380-489, builders :492-555.
The module tree reproduces the reference's state-dict keys ('stem.stem_1/conv.weight', ...,
'stage3.OSA3_1.concat.OSA3_1_concat/norm.running_var', 'stage3.OSA3_1.ese.fc.bias').

MI355X layout: NHWC fp32.  An OSA block owns ONE buffer of in_ch + 5*stage_ch channels; its producer (stem_3, the
max pool, or the previous block's eSE scale) writes channels [0,in_ch), each 3x3 conv writes its own slice and the
1x1 aggregation reads the whole buffer — torch.cat (vovnet.py:324) never happens.  FrozenBN is folded into the conv
epilogue (scale/shift), ReLU too.
"""
from collections import OrderedDict

import torch
from torch import nn

from ... import ops
from ...ops import View
from ...registry import BACKBONE_REGISTRY
from ...structures import ShapeSpec
from ..base import Backbone, FrozenBatchNorm2d, HipModule
from .fpn import FPN, LastLevelMaxPool, LastLevelP6, LastLevelP6P7

__all__ = ["VoVNet", "build_vovnet_backbone", "build_fcos_vovnet_fpn_backbone", "_STAGE_SPECS"]

# vovnet.py:30-108
_STAGE_SPECS = {
    "V-19-slim-dw-eSE": dict(stem=[64, 64, 64], stage_conv_ch=[64, 80, 96, 112], stage_out_ch=[112, 256, 384, 512],
                             layer_per_block=3, block_per_stage=[1, 1, 1, 1], eSE=True, dw=True),
    "V-19-dw-eSE": dict(stem=[64, 64, 64], stage_conv_ch=[128, 160, 192, 224], stage_out_ch=[256, 512, 768, 1024],
                        layer_per_block=3, block_per_stage=[1, 1, 1, 1], eSE=True, dw=True),
    "V-19-slim-eSE": dict(stem=[64, 64, 128], stage_conv_ch=[64, 80, 96, 112], stage_out_ch=[112, 256, 384, 512],
                          layer_per_block=3, block_per_stage=[1, 1, 1, 1], eSE=True, dw=False),
    "V-19-eSE": dict(stem=[64, 64, 128], stage_conv_ch=[128, 160, 192, 224], stage_out_ch=[256, 512, 768, 1024],
                     layer_per_block=3, block_per_stage=[1, 1, 1, 1], eSE=True, dw=False),
    "V-39-eSE": dict(stem=[64, 64, 128], stage_conv_ch=[128, 160, 192, 224], stage_out_ch=[256, 512, 768, 1024],
                     layer_per_block=5, block_per_stage=[1, 1, 2, 2], eSE=True, dw=False),
    "V-57-eSE": dict(stem=[64, 64, 128], stage_conv_ch=[128, 160, 192, 224], stage_out_ch=[256, 512, 768, 1024],
                     layer_per_block=5, block_per_stage=[1, 1, 4, 3], eSE=True, dw=False),
    "V-99-eSE": dict(stem=[64, 64, 128], stage_conv_ch=[128, 160, 192, 224], stage_out_ch=[256, 512, 768, 1024],
                     layer_per_block=5, block_per_stage=[1, 3, 9, 3], eSE=True, dw=False),
}


def _conv_bn(in_ch, out_ch, module_name, postfix, k, stride=1):
    """Parameter holders named like vovnet.py:205-236 ('<module>_<postfix>/conv', '/norm')."""
    conv = nn.Conv2d(in_ch, out_ch, kernel_size=k, stride=stride, padding=k // 2, bias=False)
    nn.init.kaiming_normal_(conv.weight)
    return [("{}_{}/conv".format(module_name, postfix), conv), ("{}_{}/norm".format(module_name, postfix), FrozenBatchNorm2d(out_ch))]


def _dw_pw_bn(ch, module_name, postfix, stride=1):
    """Parameter holders of dw_conv3x3 (vovnet.py:110-130): '/dw_conv3x3' (depth-wise, carries the stride), '/pw_conv1x1', '/pw_norm'."""
    dw = nn.Conv2d(ch, ch, kernel_size=3, stride=stride, padding=1, groups=ch, bias=False)
    pw = nn.Conv2d(ch, ch, kernel_size=1, bias=False)
    nn.init.kaiming_normal_(pw.weight)
    base = "{}_{}".format(module_name, postfix)
    return [(base + "/dw_conv3x3", dw), (base + "/pw_conv1x1", pw), (base + "/pw_norm", FrozenBatchNorm2d(ch))]


def _fold_dw(seq: nn.Module, name: str, dev, stride=1):
    """-> (tap-major dw weight on the device, PackedConv of the point-wise 1x1 with the FrozenBN folded in, stride)."""
    dw, pw, norm = getattr(seq, name + "/dw_conv3x3"), getattr(seq, name + "/pw_conv1x1"), getattr(seq, name + "/pw_norm")
    scale, shift = ops.fold_frozen_bn(norm.weight, norm.bias, norm.running_mean, norm.running_var, norm.eps)
    return ops.pack_dw_weight(dw.weight).to(dev), ops.PackedConv(pw.weight, scale, shift, dev), stride


def _dw_layer(src: View, packed, dst: View) -> None:
    """dw 3x3 into a scratch tensor, then the 1x1 + BN + ReLU into the destination slice."""
    w9c, pw, stride = packed
    ops.conv2d(ops.dwconv3x3(src, w9c, stride=stride), pw, dst, relu=True)


def _fold(seq: nn.Module, name: str):
    conv, norm = getattr(seq, name + "/conv"), getattr(seq, name + "/norm")
    scale, shift = ops.fold_frozen_bn(norm.weight, norm.bias, norm.running_mean, norm.running_var, norm.eps)
    return conv, scale, shift


class eSEModule(nn.Module):
    def __init__(self, channel):
        super().__init__()
        self.fc = nn.Conv2d(channel, channel, kernel_size=1, padding=0)


class _OSA_module(nn.Module):
    def __init__(self, in_ch, stage_ch, concat_ch, layer_per_block, module_name, identity=False, depthwise=False):
        super().__init__()
        self.identity = identity
        self.depthwise = depthwise
        self.isReduced = False
        self.in_ch, self.stage_ch, self.concat_ch, self.module_name = in_ch, stage_ch, concat_ch, module_name
        self.layers = nn.ModuleList()
        c = in_ch
        if depthwise and in_ch != stage_ch:         # vovnet.py:284-288
            self.isReduced = True
            self.conv_reduction = nn.Sequential(OrderedDict(_conv_bn(in_ch, stage_ch, module_name + "_reduction", "0", 1)))
        for i in range(layer_per_block):
            if depthwise:
                seq = nn.Sequential(OrderedDict(_dw_pw_bn(stage_ch, module_name, i)))
            else:
                seq = nn.Sequential(OrderedDict(_conv_bn(c, stage_ch, module_name, i, 3)))
            self.layers.append(seq)
            c = stage_ch
        self.cat_ch = in_ch + layer_per_block * stage_ch
        self.concat = nn.Sequential(OrderedDict(_conv_bn(self.cat_ch, concat_ch, module_name, "concat", 1)))
        self.ese = eSEModule(concat_ch)   # unconditional in the reference (vovnet.py:307,327)


class _OSA_stage(nn.Sequential):
    def __init__(self, in_ch, stage_ch, concat_ch, block_per_stage, layer_per_block, stage_num, depthwise=False):
        super().__init__()
        self.stage_num = stage_num
        name = "OSA{}_1".format(stage_num)
        self.add_module(name, _OSA_module(in_ch, stage_ch, concat_ch, layer_per_block, name, depthwise=depthwise))
        for i in range(block_per_stage - 1):
            name = "OSA{}_{}".format(stage_num, i + 2)
            self.add_module(name, _OSA_module(concat_ch, stage_ch, concat_ch, layer_per_block, name, identity=True, depthwise=depthwise))

    def blocks(self):
        return [m for m in self.children() if isinstance(m, _OSA_module)]


class VoVNet(Backbone):
    def __init__(self, cfg, input_ch, out_features=None):
        super().__init__()
        body = cfg.MODEL.VOVNET.CONV_BODY
        if body not in _STAGE_SPECS:
            raise NotImplementedError("unknown VoVNet body {}".format(body))
        if cfg.MODEL.VOVNET.NORM != "FrozenBN":
            raise NotImplementedError("MODEL.VOVNET.NORM={} (inference path folds FrozenBN)".format(cfg.MODEL.VOVNET.NORM))
        if any(cfg.MODEL.VOVNET.STAGE_WITH_DCN):
            raise NotImplementedError("deformable convs are disabled in the reference config and not built")
        assert input_ch == 3, "stem kernel reads a 3-channel image"
        spec = _STAGE_SPECS[body]
        stem_ch = spec["stem"]
        self.depthwise = spec["dw"]
        self._out_features = list(out_features)
        stem = _conv_bn(input_ch, stem_ch[0], "stem", "1", 3, 2)
        if self.depthwise:                           # vovnet.py:408-411
            assert stem_ch[0] == stem_ch[1] == stem_ch[2], "dw_conv3x3 keeps the channel count"
            stem += _dw_pw_bn(stem_ch[1], "stem", "2", 1)
            stem += _dw_pw_bn(stem_ch[2], "stem", "3", 2)
        else:
            stem += _conv_bn(stem_ch[0], stem_ch[1], "stem", "2", 3, 1)
            stem += _conv_bn(stem_ch[1], stem_ch[2], "stem", "3", 3, 2)
        self.add_module("stem", nn.Sequential(OrderedDict(stem)))
        stride = 4
        self._out_feature_strides = {"stem": stride, "stage2": stride}
        self._out_feature_channels = {"stem": stem_ch[2]}
        in_ch_list = [stem_ch[2]] + spec["stage_out_ch"][:-1]
        self.stage_names = []
        for i in range(4):
            name = "stage%d" % (i + 2)
            self.stage_names.append(name)
            self.add_module(name, _OSA_stage(in_ch_list[i], spec["stage_conv_ch"][i], spec["stage_out_ch"][i],
                                             spec["block_per_stage"][i], spec["layer_per_block"], i + 2, depthwise=self.depthwise))
            self._out_feature_channels[name] = spec["stage_out_ch"][i]
            if i != 0:
                stride *= 2
                self._out_feature_strides[name] = stride

    # -- packed weights ------------------------------------------------------------------------------------------
    def _build_packed(self, dev):
        P = {}
        conv, sc, sh = _fold(self.stem, "stem_1")
        P["stem_1"] = (conv.weight.detach().float().cpu().permute(2, 3, 1, 0).reshape(27, -1).contiguous().to(dev), sc.to(dev), sh.to(dev))
        for nm, stride in (("stem_2", 1), ("stem_3", 2)):
            if self.depthwise:
                P[nm] = _fold_dw(self.stem, nm, dev, stride)
                continue
            conv, sc, sh = _fold(self.stem, nm)
            P[nm] = ops.PackedConv(conv.weight, sc, sh, dev, stride=stride)
        for sname in self.stage_names:
            for blk in getattr(self, sname).blocks():
                mn = blk.module_name
                if blk.isReduced:
                    conv, sc, sh = _fold(blk.conv_reduction, mn + "_reduction_0")
                    P[mn + "_reduction"] = ops.PackedConv(conv.weight, sc, sh, dev)
                for i, seq in enumerate(blk.layers):
                    if blk.depthwise:
                        P["{}_{}".format(mn, i)] = _fold_dw(seq, "{}_{}".format(mn, i), dev)
                        continue
                    conv, sc, sh = _fold(seq, "{}_{}".format(mn, i))
                    P["{}_{}".format(mn, i)] = ops.PackedConv(conv.weight, sc, sh, dev)
                conv, sc, sh = _fold(blk.concat, mn + "_concat")
                P[mn + "_concat"] = ops.PackedConv(conv.weight, sc, sh, dev)
                c = blk.concat_ch
                P[mn + "_ese"] = (blk.ese.fc.weight.detach().float().reshape(c, c).contiguous().to(dev),
                                  blk.ese.fc.bias.detach().float().contiguous().to(dev))
        return P

    # -- forward ---------------------------------------------------------------------------------------------------
    def forward_views(self, x: torch.Tensor):
        """x: (N,3,H,W) float32 on the GPU -> {name: View} for the requested features (vovnet.py:471-481)."""
        if x.dim() != 4 or x.shape[1] != 3:
            raise ValueError("VoVNet expects (N,3,H,W), got {}".format(tuple(x.shape)))
        P = self.packed()
        dev = x.device
        outputs = {}
        w27, sc, sh = P["stem_1"]
        s1 = ops.stem_conv(x.float(), w27, sc, sh)
        if self.depthwise:
            s2 = View(torch.empty_like(s1.t))
            _dw_layer(s1, P["stem_2"], s2)
        else:
            s2 = ops.conv_out(s1, P["stem_2"], relu=True)
        n = x.shape[0]
        prev = None        # dense output View of the previous stage
        prev_gate = None   # eSE gate still to be applied to `prev` (folded into the next max pool)
        for sname in self.stage_names:
            blocks = getattr(self, sname).blocks()
            if sname == "stage2":
                h, w = (s2.t.shape[1] - 1) // 2 + 1, (s2.t.shape[2] - 1) // 2 + 1
            else:
                h, w = _pool_out(prev.t.shape[1]), _pool_out(prev.t.shape[2])
            cat = torch.empty((n, h, w, blocks[0].cat_ch), dtype=torch.float32, device=dev)
            inp = View(cat, 0, blocks[0].in_ch)
            if sname == "stage2":
                if "stem" in self._out_features:
                    stem_out = View(torch.empty((n, h, w, blocks[0].in_ch), dtype=torch.float32, device=dev))
                    self._stem3(s2, P["stem_3"], stem_out)
                    outputs["stem"] = stem_out
                    inp.t[..., :blocks[0].in_ch].copy_(stem_out.t)
                else:
                    self._stem3(s2, P["stem_3"], inp)
            else:
                ops.maxpool3x3s2_ceil(prev, inp, gate=prev_gate)     # vovnet.py:349-350 (+ the pending eSE scale, see below)
                prev_gate = None
            for b, blk in enumerate(blocks):
                mn = blk.module_name
                src, off = inp, blk.in_ch
                if blk.isReduced:                                              # vovnet.py:317-318
                    src = ops.conv_out(inp, P[mn + "_reduction"], relu=True)
                for i in range(len(blk.layers)):
                    dst = View(cat, off, blk.stage_ch)
                    if blk.depthwise:
                        _dw_layer(src, P["{}_{}".format(mn, i)], dst)
                    else:
                        ops.conv2d(src, P["{}_{}".format(mn, i)], dst, relu=True)
                    src, off = dst, off + blk.stage_ch
                pooled = []             # the aggregation conv leaves the eSE average pool's partial sums behind when its kernel can
                xt = ops.conv_out(View(cat), P[mn + "_concat"], relu=True, pool=pooled)     # 1x1 over the un-materialised concat
                if b + 1 < len(blocks):
                    nxt = torch.empty((n, h, w, blocks[b + 1].cat_ch), dtype=torch.float32, device=dev)
                    out = View(nxt, 0, blk.concat_ch)
                else:
                    nxt = None
                    out = View(torch.empty((n, h, w, blk.concat_ch), dtype=torch.float32, device=dev))
                fw, fb = P[mn + "_ese"]
                gate = ops.ese_gate_pooled(pooled[0], fw, fb, n, h * w) if pooled else None
                if nxt is None and not blk.identity and sname not in self._out_features and sname != self.stage_names[-1]:
                    # nobody but the next stage's max pool reads this output: max(x*g) == g*max(x) for the non-negative
                    # hsigmoid gate, so the scale pass (a full read+write of the largest map) is folded into the pool
                    prev_gate = gate if gate is not None else ops.ese_gate(xt, fw, fb)
                    out = xt
                else:
                    ops.ese(xt, fw, fb, out, identity=inp if blk.identity else None, gate=gate)   # eSE then identity add (:327-330)
                if nxt is not None:
                    cat, inp = nxt, out
            prev = out
            if sname in self._out_features:
                outputs[sname] = out
        return outputs

    def _stem3(self, s2: View, packed, dst: View) -> None:
        if self.depthwise:
            _dw_layer(s2, packed, dst)
        else:
            ops.conv2d(s2, packed, dst, relu=True)

    def forward(self, x):
        return {k: v.nchw() for k, v in self.forward_views(x).items()}


def _pool_out(n: int) -> int:
    o = -(-(n - 3) // 2) + 1
    return o - 1 if (o - 1) * 2 >= n else o


@BACKBONE_REGISTRY.register()
def build_vovnet_backbone(cfg, input_shape):
    """vovnet.py:492-501."""
    return VoVNet(cfg, input_shape.channels, out_features=cfg.MODEL.VOVNET.OUT_FEATURES)


@BACKBONE_REGISTRY.register()
def build_vovnet_fpn_backbone(cfg, input_shape: ShapeSpec):
    """vovnet.py:504-524: VoVNet + FPN with d2's LastLevelMaxPool on top (p6 = every second pixel of p5) — the backbone of the
    reference's Mask R-CNN style configs; the FCOS configs use build_fcos_vovnet_fpn_backbone below."""
    bottom_up = build_vovnet_backbone(cfg, input_shape)
    return FPN(bottom_up=bottom_up, in_features=cfg.MODEL.FPN.IN_FEATURES, out_channels=cfg.MODEL.FPN.OUT_CHANNELS, norm=cfg.MODEL.FPN.NORM,
               top_block=LastLevelMaxPool(), fuse_type=cfg.MODEL.FPN.FUSE_TYPE)


@BACKBONE_REGISTRY.register()
def build_fcos_vovnet_fpn_backbone(cfg, input_shape: ShapeSpec):
    """vovnet.py:527-555: VoVNet + FPN(in=stage3-5, out=256, norm "", fuse "sum") + P6/P7 from p5."""
    bottom_up = build_vovnet_backbone(cfg, input_shape)
    in_features = cfg.MODEL.FPN.IN_FEATURES
    out_channels = cfg.MODEL.FPN.OUT_CHANNELS
    top_levels = cfg.MODEL.FCOS.TOP_LEVELS
    if top_levels == 2:
        top_block = LastLevelP6P7(out_channels, out_channels, "p5")
    elif top_levels == 1:
        top_block = LastLevelP6(out_channels, out_channels, "p5")
    elif top_levels == 0:
        top_block = None
    else:
        raise ValueError("MODEL.FCOS.TOP_LEVELS must be 0, 1 or 2")
    return FPN(bottom_up=bottom_up, in_features=in_features, out_channels=out_channels, norm=cfg.MODEL.FPN.NORM,
               top_block=top_block, fuse_type=cfg.MODEL.FPN.FUSE_TYPE)
