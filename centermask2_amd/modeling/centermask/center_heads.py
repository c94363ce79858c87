"""CenterROIHeads (mask -> mask-IoU cascade on FCOS boxes) on HIP kernels.

Mirrors the inference half of centermask2/centermask/modeling/centermask/:
  center_heads.py:295-355,384-517  CenterROIHeads.__init__/forward/forward_with_given_boxes/_forward_mask/_forward_maskiou
  pooler.py:70-118,155-189,290-366  ROIPooler with ROIAlignV2 + "ratio" level assignment  -> ops.roi_align_ratio
  sam.py:12-97                      SpatialAttentionMaskHead                               -> MFMA convs + ops.spatial_attention_
  mask_head.py:174-216              mask_rcnn_inference (class-selected sigmoid)           -> ops.mask_predict
  maskiou_head.py:50-120            MaskIoUHead + mask_iou_inference                        -> MFMA convs / FCs + ops.mask_iou_score

ROIs are kept in a padded [image][K] layout with device-side counts, so the launch sequence is static.
The ROI feature buffer has 272 channels: [0,256) ROIAlign output, 256 = 2x2-max-pooled mask (the torch.cat of
maskiou_head.py:112 becomes a channel write), 257..271 zero padding for the 16-channel K chunks.
"""
import math
from typing import Dict, List, Optional, Tuple

import torch
from torch import nn

from ... import ops
from ...ops import View
from ...registry import ROI_HEADS_REGISTRY, ROI_MASK_HEAD_REGISTRY, ROI_MASKIOU_HEAD_REGISTRY
from ...structures import Boxes, Instances, ShapeSpec
from ..base import HipModule, NormConv2d, fold_norm, get_norm

__all__ = ["CenterROIHeads", "ROIPooler", "SpatialAttentionMaskHead", "MaskIoUHead", "build_mask_head", "build_maskiou_head"]


class ROIPooler(nn.Module):
    """pooler.py:192-288 constructor surface.  ROIAlignV2 (the reference config's type) and ROIAlign v1, level assignment by
    "ratio" (the reference recipe) or by "area" (FPN Eqn.(1), pooler.py:121-152); ROIPool / ROIAlignRotated are out of scope."""

    def __init__(self, output_size, scales, sampling_ratio, pooler_type, canonical_box_size=224, canonical_level=4, assign_crit="area"):
        super().__init__()
        if isinstance(output_size, (tuple, list)):
            assert output_size[0] == output_size[1]
            output_size = output_size[0]
        if pooler_type not in ("ROIAlignV2", "ROIAlign"):
            raise NotImplementedError("pooler_type {} (ROIAlignV2 / ROIAlign are built; ROIPool and ROIAlignRotated are out of scope)".format(pooler_type))
        if assign_crit not in ("ratio", "area"):
            raise ValueError("unknown ASSIGN_CRITERION {}".format(assign_crit))
        self.aligned = pooler_type == "ROIAlignV2"
        self.assign_crit = assign_crit
        assert canonical_box_size > 0
        self.canonical_box_size, self.canonical_level = canonical_box_size, canonical_level
        self.output_size = output_size
        self.scales = tuple(scales)
        self.sampling_ratio = sampling_ratio
        min_level, max_level = -math.log2(scales[0]), -math.log2(scales[-1])
        assert math.isclose(min_level, int(min_level)) and math.isclose(max_level, int(max_level)), "Featuremap stride is not power of 2!"
        self.min_level, self.max_level = int(min_level), int(max_level)
        assert len(scales) == self.max_level - self.min_level + 1, "[ROIPooler] Sizes of input featuremaps do not form a pyramid!"


class SpatialAttention(nn.Module):
    def __init__(self, kernel_size=3):
        super().__init__()
        assert kernel_size == 3, "only the 3x3 spatial attention of the reference config is built"
        self.conv = nn.Conv2d(2, 1, kernel_size, padding=1, bias=False)


@ROI_MASK_HEAD_REGISTRY.register()
class SpatialAttentionMaskHead(HipModule):
    """sam.py:31-97."""

    def __init__(self, cfg, input_shape: ShapeSpec):
        super().__init__()
        num_classes = cfg.MODEL.ROI_HEADS.NUM_CLASSES
        conv_dims = cfg.MODEL.ROI_MASK_HEAD.CONV_DIM
        self.norm = cfg.MODEL.ROI_MASK_HEAD.NORM           # sam.py:53: "" in the reference recipe; "GN" / "FrozenBN" / "BN" as d2's get_norm
        num_conv = cfg.MODEL.ROI_MASK_HEAD.NUM_CONV
        self.cls_agnostic_mask = cfg.MODEL.ROI_MASK_HEAD.CLS_AGNOSTIC_MASK
        self.num_conv = num_conv
        self.conv_dims = conv_dims
        for k in range(num_conv):
            conv = NormConv2d(input_shape.channels if k == 0 else conv_dims, conv_dims, kernel_size=3, stride=1, padding=1, bias=not self.norm,
                              norm=get_norm(self.norm, conv_dims))                     # sam.py:58-70
            nn.init.kaiming_normal_(conv.weight, mode="fan_out", nonlinearity="relu")
            if conv.bias is not None:
                nn.init.constant_(conv.bias, 0)
            self.add_module("mask_fcn{}".format(k + 1), conv)
        self.spatialAtt = SpatialAttention()
        self.deconv = nn.ConvTranspose2d(conv_dims if num_conv > 0 else input_shape.channels, conv_dims, kernel_size=2, stride=2, padding=0)
        self.predictor = nn.Conv2d(conv_dims, 1 if self.cls_agnostic_mask else num_classes, kernel_size=1, stride=1, padding=0)
        nn.init.normal_(self.predictor.weight, std=0.001)
        nn.init.constant_(self.predictor.bias, 0)

    def _build_packed(self, dev):
        P = {"convs": []}
        for k in range(self.num_conv):
            c = getattr(self, "mask_fcn{}".format(k + 1))
            scale, shift, gn = fold_norm(c)
            P["convs"].append((ops.PackedConv(c.weight, scale, shift, dev), None if gn is None else (gn[0].contiguous().to(dev), gn[1].contiguous().to(dev), gn[2], gn[3])))
        P["sam_w"] = self.spatialAtt.conv.weight.detach().float().reshape(-1).contiguous().to(dev)      # [ci][kh][kw]
        # ConvTranspose2d k2 s2: out[2h+dh, 2w+dw, co] = sum_ci x[h,w,ci] W[ci,co,dh,dw] + b[co]  ==  a 1x1 conv with
        # 4*C outputs ordered (dh,dw,co)
        w = self.deconv.weight.detach().float().cpu()                     # (Cin, Cout, 2, 2)
        cin, cout = w.shape[0], w.shape[1]
        w1 = w.permute(2, 3, 1, 0).reshape(4 * cout, cin, 1, 1)
        b1 = self.deconv.bias.detach().float().cpu().repeat(4)
        P["deconv"] = ops.PackedConv(w1, None, b1, dev)
        P["pred_w"] = self.predictor.weight.detach().float().reshape(self.predictor.out_channels, -1).contiguous().to(dev)
        P["pred_b"] = self.predictor.bias.detach().float().contiguous().to(dev)
        P["pred_conv"] = ops.PackedConv(self.predictor.weight, None, self.predictor.bias, dev)      # all classes (tests / forward())
        return P

    def features(self, x: View, counts: torch.Tensor, topk: int) -> torch.Tensor:
        """conv x4 -> spatial attention -> relu(deconv): returns (R,S,S,4*C) with the 2x2 sub-pixels (dh,dw)-major."""
        P = self.packed()
        for pc, gn in P["convs"]:
            x = ops.conv_out(x, pc, relu=gn is None)
            if gn is not None:                                   # conv -> GroupNorm -> ReLU (sam.py:58-70 with NORM "GN")
                ops.groupnorm_relu_(x.t, gn[0], gn[1], gn[3], gn[2])
        ops.spatial_attention_(x.t, P["sam_w"], counts, topk)
        return ops.conv_out(x, P["deconv"], relu=True).t

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """Reference signature sam.py:92-97: (M,C,14,14) -> mask logits (M,classes,28,28) for every class."""
        P = self.packed()
        xv = ops.as_view(x)
        m = xv.t.shape[0]
        counts = torch.full((1,), m, dtype=torch.int32, device=x.device)
        dec = self.features(xv, counts, m)                                   # (M,S,S,4C)
        s, c = dec.shape[1], dec.shape[3] // 4
        lg = ops.conv_out(View(dec.reshape(m, s, s * 4, c)), P["pred_conv"]).t     # (M,S,S*4,K)
        k = lg.shape[3]
        return lg.reshape(m, s, s, 2, 2, k).permute(0, 5, 1, 3, 2, 4).reshape(m, k, 2 * s, 2 * s)


@ROI_MASKIOU_HEAD_REGISTRY.register()
class MaskIoUHead(HipModule):
    """maskiou_head.py:63-120."""

    def __init__(self, cfg, input_shape: ShapeSpec):
        super().__init__()
        num_classes = cfg.MODEL.ROI_HEADS.NUM_CLASSES
        conv_dims = cfg.MODEL.ROI_MASKIOU_HEAD.CONV_DIM
        num_conv = cfg.MODEL.ROI_MASKIOU_HEAD.NUM_CONV
        input_channels = input_shape.channels + 1
        self.resolution = input_shape.width // 2
        self.num_conv = num_conv
        self.conv_dims = conv_dims
        for k in range(num_conv):
            conv = nn.Conv2d(input_channels if k == 0 else conv_dims, conv_dims, kernel_size=3, stride=2 if (k + 1) == num_conv else 1, padding=1)
            nn.init.kaiming_normal_(conv.weight, mode="fan_out", nonlinearity="relu")
            nn.init.constant_(conv.bias, 0)
            self.add_module("maskiou_fcn{}".format(k + 1), conv)
        self.maskiou_fc1 = nn.Linear(conv_dims * self.resolution ** 2, 1024)
        self.maskiou_fc2 = nn.Linear(1024, 1024)
        self.maskiou = nn.Linear(1024, num_classes)
        nn.init.normal_(self.maskiou.weight, mean=0, std=0.01)
        nn.init.constant_(self.maskiou.bias, 0)

    def _build_packed(self, dev):
        P = {"convs": []}
        for k in range(self.num_conv):
            c = getattr(self, "maskiou_fcn{}".format(k + 1))
            P["convs"].append(ops.PackedConv(c.weight, None, c.bias, dev, stride=c.stride[0]))
        # torch.flatten(x, 1) of NCHW (maskiou_head.py:115) -> our rows are NHWC: permute fc1's columns (c,h,w)->(h,w,c)
        r, c = self.resolution, self.conv_dims
        w1 = self.maskiou_fc1.weight.detach().float().cpu().reshape(-1, c, r, r).permute(0, 2, 3, 1).reshape(-1, c * r * r)
        P["fc1"] = ops.PackedConv(w1, None, self.maskiou_fc1.bias, dev)
        P["fc2"] = ops.PackedConv(self.maskiou_fc2.weight, None, self.maskiou_fc2.bias, dev)
        P["fc3"] = ops.PackedConv(self.maskiou.weight, None, self.maskiou.bias, dev)
        return P

    def forward_views(self, roi: View) -> torch.Tensor:
        """roi: (R,14,14,272) view over [features | pooled mask | 0...] -> (R, num_classes) predicted mask IoU."""
        P = self.packed()
        x = roi
        for pc in P["convs"]:
            x = ops.conv_out(x, pc, relu=True)
        flat = x.t.reshape(x.t.shape[0], -1)
        h = ops.linear(flat, P["fc1"], relu=True)
        h = ops.linear(h, P["fc2"], relu=True)
        return ops.linear(h, P["fc3"])

    def forward(self, x: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
        """Reference signature maskiou_head.py:107-120: x (M,C,14,14), mask (M,1,28,28)."""
        xv = ops.as_view(x)
        m, s, _, c = xv.t.shape
        cpad = (c + 1 + 15) // 16 * 16
        roi = torch.empty((m, s, s, cpad), dtype=torch.float32, device=x.device)
        roi[..., :c] = xv.t
        ops.mask_pool_concat_(mask.contiguous().reshape(m, 2 * s, 2 * s), roi, c)
        return self.forward_views(View(roi))


def build_mask_head(cfg, input_shape):
    return ROI_MASK_HEAD_REGISTRY.get(cfg.MODEL.ROI_MASK_HEAD.NAME)(cfg, input_shape)


def build_maskiou_head(cfg, input_shape):
    return ROI_MASKIOU_HEAD_REGISTRY.get(cfg.MODEL.ROI_MASKIOU_HEAD.NAME)(cfg, input_shape)


def lazy_batch_of(instances: List[Instances]):
    """The LazyBatch behind `instances` if they are exactly the (unmodified) list one FCOS.forward call returned, else None."""
    first = instances[0]
    if not hasattr(first, "lazy_batch"):
        return None
    batch = first.lazy_batch()[0]
    if batch is None or len(instances) != len(batch.image_sizes):
        return None
    for i, it in enumerate(instances):
        if not hasattr(it, "lazy_batch") or it.lazy_batch()[0] is not batch or it.lazy_batch()[1] != i:
            return None
    return batch


def padded_from_instances(instances: List[Instances], device) -> Tuple[dict, int]:
    """Build the padded [image][K] device buffers from foreign Instances (pred_boxes, pred_classes[, scores])."""
    n = len(instances)
    k = max(1, max(len(it) for it in instances))
    det = dict(box=torch.zeros((n, k, 4), dtype=torch.float32, device=device), score=torch.zeros((n, k), dtype=torch.float32, device=device),
               cls=torch.zeros((n, k), dtype=torch.int64, device=device), loc=torch.zeros((n, k, 2), dtype=torch.float32, device=device),
               counts=torch.tensor([len(it) for it in instances], dtype=torch.int32, device=device))
    for i, it in enumerate(instances):
        m = len(it)
        if m == 0:
            continue
        det["box"][i, :m] = it.pred_boxes.tensor.to(device).float()
        det["cls"][i, :m] = it.pred_classes.to(device).long()
        if it.has("scores"):
            det["score"][i, :m] = it.scores.to(device).float()
        if it.has("locations"):
            det["loc"][i, :m] = it.locations.to(device).float()
    return det, k


@ROI_HEADS_REGISTRY.register()
class CenterROIHeads(HipModule):
    def __init__(self, cfg, input_shape: Dict[str, ShapeSpec]):
        super().__init__()
        self.in_features = cfg.MODEL.ROI_HEADS.IN_FEATURES
        self.num_classes = cfg.MODEL.ROI_HEADS.NUM_CLASSES
        self.feature_strides = {k: v.stride for k, v in input_shape.items()}
        self.feature_channels = {k: v.channels for k, v in input_shape.items()}
        self._init_mask_head(cfg)
        self._init_mask_iou_head(cfg)
        self.keypoint_on = cfg.MODEL.KEYPOINT_ON
        if self.keypoint_on:
            raise NotImplementedError("MODEL.KEYPOINT_ON (False in the reference config)")

    def _build_packed(self, dev):
        return {}

    def _init_mask_head(self, cfg):            # center_heads.py:315-339
        self.mask_on = cfg.MODEL.MASK_ON
        if not self.mask_on:
            return
        pooler_resolution = cfg.MODEL.ROI_MASK_HEAD.POOLER_RESOLUTION
        pooler_scales = tuple(1.0 / self.feature_strides[k] for k in self.in_features)
        in_channels = [self.feature_channels[f] for f in self.in_features][0]
        self.mask_pooler = ROIPooler(output_size=pooler_resolution, scales=pooler_scales,
                                     sampling_ratio=cfg.MODEL.ROI_MASK_HEAD.POOLER_SAMPLING_RATIO,
                                     pooler_type=cfg.MODEL.ROI_MASK_HEAD.POOLER_TYPE, assign_crit=cfg.MODEL.ROI_MASK_HEAD.ASSIGN_CRITERION)
        self.mask_head = build_mask_head(cfg, ShapeSpec(channels=in_channels, width=pooler_resolution, height=pooler_resolution))

    def _init_mask_iou_head(self, cfg):        # center_heads.py:342-355
        self.maskiou_on = cfg.MODEL.MASKIOU_ON
        if not self.maskiou_on:
            return
        in_channels = cfg.MODEL.ROI_MASK_HEAD.CONV_DIM
        pooler_resolution = cfg.MODEL.ROI_MASK_HEAD.POOLER_RESOLUTION
        self.maskiou_head = build_maskiou_head(cfg, ShapeSpec(channels=in_channels, width=pooler_resolution, height=pooler_resolution))

    def _img_area(self, image_sizes, dev):
        key = (tuple(tuple(hw) for hw in image_sizes), str(dev))
        cache = self.__dict__.setdefault("_img_area_cache", {})
        if key not in cache:     # built once per size tuple: no H2D copy inside a captured graph
            cache[key] = torch.tensor([float(hw[0] * hw[1]) for hw in image_sizes], dtype=torch.float32, device=dev)
        return cache[key]

    # -- device-only core ----------------------------------------------------------------------------------------
    def forward_padded(self, features, det: dict, image_sizes, want=()) -> dict:
        """det: padded detections (box (N,K,4), score, cls int64, counts int32).  Adds pred_masks (N,K,1,28,28) and
        mask_scores (N,K).  `want` may name intermediates to keep: 'roi_feat', 'levels', 'mask_logits', 'maskiou'."""
        if not self.mask_on:
            return det
        feats = [ops.as_view(features[f]) for f in self.in_features]
        dev = feats[0].t.device
        n, k = det["box"].shape[0], det["box"].shape[1]
        r = n * k
        pool = self.mask_pooler
        c = feats[0].c
        cpad = (c + 1 + 15) // 16 * 16
        s = pool.output_size
        roi = torch.empty((r, s, s, cpad), dtype=torch.float32, device=dev)
        img_area = self._img_area(image_sizes, dev)                                                               # pooler.py:70-77
        levels = ops.roi_align_ratio(feats, pool.scales, det["box"], det["counts"], img_area, s, pool.sampling_ratio, roi, pool.min_level,
                                     aligned=pool.aligned, assign_by_area=pool.assign_crit == "area",
                                     canonical_box_size=pool.canonical_box_size, canonical_level=pool.canonical_level)
        dec = self.mask_head.features(View(roi, 0, c), det["counts"], k)
        P = self.mask_head.packed()
        cls_flat = det["cls"].reshape(-1)
        if self.mask_head.cls_agnostic_mask:
            cls_flat = torch.zeros_like(cls_flat)
        res = ops.mask_predict(dec, P["pred_w"], P["pred_b"], cls_flat, det["counts"], k, want_logits="mask_logits" in want)
        masks, sel_logits = res if "mask_logits" in want else (res, None)
        out = dict(det)
        out["pred_masks"] = masks.reshape(n, k, 1, 2 * s, 2 * s)
        if self.maskiou_on:
            ops.mask_pool_concat_(masks.reshape(r, 2 * s, 2 * s), roi, c)
            iou = self.maskiou_head.forward_views(View(roi))
            out["mask_scores"] = ops.mask_iou_score(iou, det["score"].reshape(-1), det["cls"].reshape(-1)).reshape(n, k)
            if "maskiou" in want:
                out["maskiou"] = iou
        if "roi_feat" in want:
            out["roi_feat"] = roi[..., :c].clone()
        if "levels" in want:
            out["levels"] = levels
        if sel_logits is not None:
            out["mask_logits_selected"] = sel_logits
        return out

    # -- reference API ---------------------------------------------------------------------------------------------
    def forward(self, images, features, proposals: List[Instances], targets=None):
        """center_heads.py:384-411 (inference branch)."""
        del images
        if self.training:
            raise NotImplementedError("training is out of scope of the MI355X inference path")
        return self.forward_with_given_boxes(features, proposals), {}

    def forward_with_given_boxes(self, features, instances: List[Instances]) -> List[Instances]:
        """center_heads.py:413-444: adds pred_masks and mask_scores in place; with zero boxes overall the reference returns
        without mask_scores (center_heads.py:513-514) — here an empty tensor is attached instead."""
        assert not self.training
        dev = next(iter(features.values())).device
        batch = lazy_batch_of(instances)
        if batch is not None and batch.counts is None:
            # the proposals are FCOS's own, still on the device and unread: enqueue the ROI heads on the padded buffers and leave the
            # result with the batch — the Instances gain pred_masks / mask_scores when their fields are first read (no host sync here)
            sizes = [it.image_size for it in instances]
            out = self.forward_padded(features, batch.det, sizes)
            batch.attach_roi(out, lambda det2: self.forward_padded(features, det2, sizes))
            return instances
        assert instances[0].has("pred_boxes") and instances[0].has("pred_classes")
        if batch is not None:
            det, k = batch.det, batch.det["box"].shape[1]
        else:
            det, k = padded_from_instances(instances, dev)
        out = self.forward_padded(features, det, [it.image_size for it in instances])
        for i, it in enumerate(instances):
            m = len(it)
            if self.mask_on:
                it.pred_masks = out["pred_masks"][i, :m]
            if self.mask_on and self.maskiou_on:
                it.mask_scores = out["mask_scores"][i, :m]
        return instances
