"""
ORACLE — test infrastructure only.  Only tests/, __graft_entry__.smoke() and bench.py's
`cpu_baseline` leg may import this module; the product path (centermask2_amd/) never does.

CPU restatement, in plain PyTorch fp32 ops (NCHW) plus the C kernels of oracle_ops.c, of the
CenterMask2 inference path of Zeng-Yan/centermask2 (paths below are relative to /root/reference):

    VoVNetV2-eSE backbone   centermask2/centermask/modeling/backbone/vovnet.py:205-260,263-376,380-489
    FPN + P6/P7             detectron2 FPN (source absent, built at vovnet.py:547-554) + backbone/fpn.py:17-35
    FCOS head               modeling/fcos/fcos.py:19-25,120-144,147-240
    FCOS decode/NMS/top-k   modeling/fcos/fcos_outputs.py:372-495, layers/ml_nms.py:65-98
    ROI pooler (ratio)      modeling/centermask/pooler.py:70-118,155-189,290-366
    SAG-Mask head           modeling/centermask/sam.py:12-97, mask_head.py:174-216
    MaskIoU head            modeling/centermask/maskiou_head.py:50-120, center_heads.py:492-517

It consumes a state dict in the reference's key names (full-model prefixes `backbone.bottom_up.`,
`backbone.`, `proposal_generator.fcos_head.`, `roi_heads.mask_head.`, `roi_heads.maskiou_head.`).

Pinning: the reference has no tests or golden vectors.  tests/golden/make_golden.py imports the
reference's own modules in the build container (third-party names it needs — detectron2, fvcore —
are absent and are stubbed there) and checks this file against them module by module; the outputs are
committed under tests/golden/.  The detectron2/torchvision pieces (FrozenBN, FPN wiring, ROIAlign,
batched_nms, Instances) have no source in the reference tree and stay "parity unpinned".
"""
import ctypes
import math
import os
import sys
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "_build", "liboracle_ops.so")
        if not os.path.exists(path):
            import subprocess
            subprocess.check_call(["make", "-s", "-C", _HERE])
        lib = ctypes.CDLL(path)
        lib.oracle_nms.restype = ctypes.c_int64
        lib.oracle_nms.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p]
        lib.oracle_roi_align.restype = None
        lib.oracle_roi_align.argtypes = [
            ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64,
            ctypes.c_float, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p,
        ]
        _LIB = lib
    return _LIB


# --------------------------------------------------------------------------------------
# VoVNetV2 (vovnet.py)
# --------------------------------------------------------------------------------------
# vovnet.py:30-108
STAGE_SPECS = {
    "V-19-slim-dw-eSE": dict(stem=[64, 64, 64], stage_conv_ch=[64, 80, 96, 112], stage_out_ch=[112, 256, 384, 512],
                             layer_per_block=3, block_per_stage=[1, 1, 1, 1], dw=True),
    "V-19-dw-eSE": dict(stem=[64, 64, 64], stage_conv_ch=[128, 160, 192, 224], stage_out_ch=[256, 512, 768, 1024],
                        layer_per_block=3, block_per_stage=[1, 1, 1, 1], dw=True),
    "V-19-slim-eSE": dict(stem=[64, 64, 128], stage_conv_ch=[64, 80, 96, 112], stage_out_ch=[112, 256, 384, 512],
                          layer_per_block=3, block_per_stage=[1, 1, 1, 1]),
    "V-19-eSE": dict(stem=[64, 64, 128], stage_conv_ch=[128, 160, 192, 224], stage_out_ch=[256, 512, 768, 1024],
                     layer_per_block=3, block_per_stage=[1, 1, 1, 1]),
    "V-39-eSE": dict(stem=[64, 64, 128], stage_conv_ch=[128, 160, 192, 224], stage_out_ch=[256, 512, 768, 1024],
                     layer_per_block=5, block_per_stage=[1, 1, 2, 2]),
    "V-57-eSE": dict(stem=[64, 64, 128], stage_conv_ch=[128, 160, 192, 224], stage_out_ch=[256, 512, 768, 1024],
                     layer_per_block=5, block_per_stage=[1, 1, 4, 3]),
    "V-99-eSE": dict(stem=[64, 64, 128], stage_conv_ch=[128, 160, 192, 224], stage_out_ch=[256, 512, 768, 1024],
                     layer_per_block=5, block_per_stage=[1, 3, 9, 3]),
}
FROZEN_BN_EPS = 1e-5  # detectron2 FrozenBatchNorm2d default (source absent)


def frozen_bn(x: torch.Tensor, sd: Dict[str, torch.Tensor], prefix: str) -> torch.Tensor:
    """d2 FrozenBatchNorm2d in eval without grad = F.batch_norm(training=False) (source absent)."""
    return F.batch_norm(x, sd[prefix + "running_mean"], sd[prefix + "running_var"], sd[prefix + "weight"],
                        sd[prefix + "bias"], training=False, eps=FROZEN_BN_EPS)


def conv_bn_relu(x, sd, prefix: str, name: str, stride: int, k: int) -> torch.Tensor:
    """conv3x3 (vovnet.py:205-219) / conv1x1 (:222-236): conv(no bias) -> FrozenBN -> ReLU.
    Module names contain '/': '<module>_<postfix>/conv', '/norm'."""
    x = F.conv2d(x, sd[prefix + name + "/conv.weight"], None, stride=stride, padding=k // 2)
    x = frozen_bn(x, sd, prefix + name + "/norm.")
    return F.relu(x)


def dw_conv_bn_relu(x, sd, prefix: str, name: str, stride: int) -> torch.Tensor:
    """dw_conv3x3 (vovnet.py:110-130): depth-wise 3x3 (groups = channels, the stride sits here) -> point-wise 1x1 ->
    FrozenBN -> ReLU; nothing between the two convs."""
    w = sd[prefix + name + "/dw_conv3x3.weight"]
    x = F.conv2d(x, w, None, stride=stride, padding=1, groups=w.shape[0])
    x = F.conv2d(x, sd[prefix + name + "/pw_conv1x1.weight"], None)
    x = frozen_bn(x, sd, prefix + name + "/pw_norm.")
    return F.relu(x)


def ese_module(x, sd, prefix: str) -> torch.Tensor:
    """eSEModule.forward vovnet.py:255-260 with Hsigmoid :243-244: x * relu6(fc(avgpool(x)) + 3) / 6."""
    g = F.adaptive_avg_pool2d(x, 1)
    g = F.conv2d(g, sd[prefix + "fc.weight"], sd[prefix + "fc.bias"])
    g = F.relu6(g + 3.0) / 6.0
    return x * g


def osa_module(x, sd, prefix: str, module_name: str, layers: int, identity: bool, depthwise: bool = False) -> torch.Tensor:
    """_OSA_module.forward vovnet.py:310-332.  eSE is unconditional (:307,:327); identity add after eSE (:329-330).
    Depth-wise bodies: a 1x1 'conv_reduction' to stage_ch first when in_ch != stage_ch (:284-288, :317-318); the
    concat still starts with the un-reduced input (:315)."""
    identity_feat = x
    output = [x]
    if depthwise and (prefix + "conv_reduction.{}_reduction_0/conv.weight".format(module_name)) in sd:
        x = conv_bn_relu(x, sd, prefix + "conv_reduction.", module_name + "_reduction_0", 1, 1)
    for i in range(layers):
        if depthwise:
            x = dw_conv_bn_relu(x, sd, prefix + "layers.{}.".format(i), "{}_{}".format(module_name, i), 1)
        else:
            x = conv_bn_relu(x, sd, prefix + "layers.{}.".format(i), "{}_{}".format(module_name, i), 1, 3)
        output.append(x)
    x = torch.cat(output, dim=1)
    xt = conv_bn_relu(x, sd, prefix + "concat.", module_name + "_concat", 1, 1)
    xt = ese_module(xt, sd, prefix + "ese.")
    if identity:
        xt = xt + identity_feat
    return xt


def vovnet_forward(sd, x: torch.Tensor, conv_body: str = "V-39-eSE",
                   out_features: Sequence[str] = ("stage3", "stage4", "stage5"),
                   prefix: str = "backbone.bottom_up.") -> Dict[str, torch.Tensor]:
    """VoVNet.forward vovnet.py:471-481; stem :409-411 (strides 2,1,2); stages :335-376 with
    MaxPool2d(3, 2, ceil_mode=True) in front of stages 3-5 (:349-350)."""
    spec = STAGE_SPECS[conv_body]
    dw = spec.get("dw", False)
    outputs = {}
    x = conv_bn_relu(x, sd, prefix + "stem.", "stem_1", 2, 3)
    if dw:   # vovnet.py:408-411: stem_2/stem_3 are dw_conv3x3 in the depth-wise bodies
        x = dw_conv_bn_relu(x, sd, prefix + "stem.", "stem_2", 1)
        x = dw_conv_bn_relu(x, sd, prefix + "stem.", "stem_3", 2)
    else:
        x = conv_bn_relu(x, sd, prefix + "stem.", "stem_2", 1, 3)
        x = conv_bn_relu(x, sd, prefix + "stem.", "stem_3", 2, 3)
    if "stem" in out_features:
        outputs["stem"] = x
    for si in range(4):
        stage_num = si + 2
        name = "stage{}".format(stage_num)
        if stage_num != 2:
            x = F.max_pool2d(x, kernel_size=3, stride=2, ceil_mode=True)
        for b in range(spec["block_per_stage"][si]):
            module_name = "OSA{}_{}".format(stage_num, b + 1)
            x = osa_module(x, sd, prefix + "{}.{}.".format(name, module_name), module_name,
                           spec["layer_per_block"], identity=(b > 0), depthwise=dw)
        if name in out_features:
            outputs[name] = x
    return outputs


# --------------------------------------------------------------------------------------
# FPN (detectron2, source absent; built at vovnet.py:547-554) + LastLevelP6P7 (fpn.py:17-35)
# --------------------------------------------------------------------------------------
def fpn_forward(sd, bottom_up: Dict[str, torch.Tensor], in_features=("stage3", "stage4", "stage5"),
                strides=(8, 16, 32), prefix: str = "backbone.", top_levels: int = 2) -> Dict[str, torch.Tensor]:
    """p5 = out5(lat5(C5)); p4 = out4(lat4(C4) + up2(lat5(C5))); p3 likewise from the pre-output sum
    (fuse_type "sum", norm ""), then p6 = conv3x3s2(p5), p7 = conv3x3s2(relu(p6)) (fpn.py:32-35;
    the top block is fed from the p5 *result* because in_feature="p5", vovnet.py:541)."""
    stages = [int(math.log2(s)) for s in strides]
    results = {}
    prev = None
    for idx in range(len(in_features) - 1, -1, -1):
        st = stages[idx]
        lat = F.conv2d(bottom_up[in_features[idx]], sd[prefix + "fpn_lateral{}.weight".format(st)],
                       sd[prefix + "fpn_lateral{}.bias".format(st)])
        if prev is not None:
            lat = lat + F.interpolate(prev, scale_factor=2.0, mode="nearest")
        prev = lat
        results["p{}".format(st)] = F.conv2d(prev, sd[prefix + "fpn_output{}.weight".format(st)],
                                             sd[prefix + "fpn_output{}.bias".format(st)], padding=1)
    top = stages[-1]
    if top_levels >= 1:
        p6 = F.conv2d(results["p{}".format(top)], sd[prefix + "top_block.p6.weight"], sd[prefix + "top_block.p6.bias"],
                      stride=2, padding=1)
        results["p{}".format(top + 1)] = p6
        if top_levels == 2:
            results["p{}".format(top + 2)] = F.conv2d(F.relu(p6), sd[prefix + "top_block.p7.weight"],
                                                      sd[prefix + "top_block.p7.bias"], stride=2, padding=1)
    return {k: results[k] for k in sorted(results, key=lambda s: int(s[1:]))}


def backbone_forward(sd, x, conv_body="V-39-eSE") -> Dict[str, torch.Tensor]:
    """build_fcos_vovnet_fpn_backbone vovnet.py:527-555: VoVNet -> FPN(+P6P7)."""
    return fpn_forward(sd, vovnet_forward(sd, x, conv_body))


# --------------------------------------------------------------------------------------
# FCOS head (fcos.py)
# --------------------------------------------------------------------------------------
def fcos_tower(x, sd, prefix: str, num_convs: int, norm: Optional[str] = "GN") -> torch.Tensor:
    """[conv3x3(bias) -> GroupNorm(32) -> ReLU] x num_convs, fcos.py:169-186.  With GN the Sequential
    indices are conv 3k, GN 3k+1; without norm conv 2k."""
    step = 3 if norm == "GN" else 2
    for k in range(num_convs):
        x = F.conv2d(x, sd[prefix + "{}.weight".format(step * k)], sd[prefix + "{}.bias".format(step * k)], padding=1)
        if norm == "GN":
            x = F.group_norm(x, 32, sd[prefix + "{}.weight".format(step * k + 1)],
                             sd[prefix + "{}.bias".format(step * k + 1)], eps=1e-5)
        x = F.relu(x)
    return x


def fcos_head_forward(sd, features: List[torch.Tensor], prefix: str = "proposal_generator.fcos_head.",
                      num_cls_convs=4, num_box_convs=4, norm="GN", use_scale=True):
    """FCOSHead.forward fcos.py:222-240: share tower is empty (NUM_SHARE_CONVS 0), centerness comes
    from the *bbox* tower (:233), reg = relu(scale_l * bbox_pred(bbox_tower)) (:234-238)."""
    logits, bbox_reg, ctrness = [], [], []
    for l, feature in enumerate(features):
        cls_t = fcos_tower(feature, sd, prefix + "cls_tower.", num_cls_convs, norm)
        box_t = fcos_tower(feature, sd, prefix + "bbox_tower.", num_box_convs, norm)
        logits.append(F.conv2d(cls_t, sd[prefix + "cls_logits.weight"], sd[prefix + "cls_logits.bias"], padding=1))
        ctrness.append(F.conv2d(box_t, sd[prefix + "ctrness.weight"], sd[prefix + "ctrness.bias"], padding=1))
        reg = F.conv2d(box_t, sd[prefix + "bbox_pred.weight"], sd[prefix + "bbox_pred.bias"], padding=1)
        if use_scale:
            reg = reg * sd[prefix + "scales.{}.scale".format(l)]
        bbox_reg.append(F.relu(reg))
    return logits, bbox_reg, ctrness


def compute_locations_per_level(h: int, w: int, stride: int) -> torch.Tensor:
    """fcos.py:131-144: (x, y) = (j*stride + stride//2, i*stride + stride//2), row-major, float32."""
    shifts_x = torch.arange(0, w * stride, step=stride, dtype=torch.float32)
    shifts_y = torch.arange(0, h * stride, step=stride, dtype=torch.float32)
    shift_y, shift_x = torch.meshgrid(shifts_y, shifts_x, indexing="ij")
    return torch.stack((shift_x.reshape(-1), shift_y.reshape(-1)), dim=1) + stride // 2


# --------------------------------------------------------------------------------------
# FCOS decode / NMS / top-k (fcos_outputs.py)
# --------------------------------------------------------------------------------------
def fcos_single_level(locations, box_cls, reg_pred, ctrness, pre_nms_thresh=0.05, thresh_with_ctr=False):
    """forward_for_single_feature_map fcos_outputs.py:396-466 for every image of the batch.
    Returns per image a dict(boxes, scores, classes, locations, loc_index).  No pre-NMS top-k (the
    fork comments it out, :444-449); the 1000x80 zero pad (:426) is inert for nonzero and is omitted."""
    N, C, H, W = box_cls.shape
    box_cls = box_cls.permute(0, 2, 3, 1).reshape(N, -1, C).sigmoid()
    box_regression = reg_pred.permute(0, 2, 3, 1).reshape(N, -1, 4)
    ctr = ctrness.permute(0, 2, 3, 1).reshape(N, -1).sigmoid()
    if thresh_with_ctr:
        box_cls = box_cls * ctr[:, :, None]
    candidate_inds = box_cls > pre_nms_thresh
    if not thresh_with_ctr:
        box_cls = box_cls * ctr[:, :, None]
    results = []
    for i in range(N):
        nz = torch.nonzero(candidate_inds[i])          # row-major: location-major, class-minor
        per_box_loc, per_class = nz[:, 0], nz[:, 1]
        per_box_cls = box_cls[i].reshape(-1)[C * per_box_loc + per_class]
        per_reg = box_regression[i][per_box_loc]
        per_locations = locations[per_box_loc]
        det = torch.stack([per_locations[:, 0] - per_reg[:, 0], per_locations[:, 1] - per_reg[:, 1],
                           per_locations[:, 0] + per_reg[:, 2], per_locations[:, 1] + per_reg[:, 3]], dim=1)
        results.append(dict(boxes=det, scores=torch.sqrt(per_box_cls), classes=per_class, locations=per_locations))
    return results


def batched_nms(boxes: torch.Tensor, scores: torch.Tensor, idxs: torch.Tensor, iou_threshold: float) -> torch.Tensor:
    """d2 batched_nms -> torchvision 0.9 batched_nms (sources absent; layers/ml_nms.py:93).
    < 40000 boxes: coordinate trick `boxes + idxs * (boxes.max() + 1)` then one nms;
    >= 40000 boxes (d2 branch): nms per class, kept indices re-sorted by score.
    Sort order: descending score, ties by ascending index (a *stable* order; the reference ops leave
    ties unspecified).  Returns kept indices in that order."""
    n = boxes.shape[0]
    if n == 0:
        return torch.empty((0,), dtype=torch.int64)
    if n < 40000:
        max_coordinate = boxes.max()
        offsets = idxs.to(boxes) * (max_coordinate + torch.tensor(1).to(boxes))
        return nms(boxes + offsets[:, None], scores, iou_threshold)
    result_mask = torch.zeros(n, dtype=torch.bool)
    for cid in torch.unique(idxs).tolist():
        mask = (idxs == cid).nonzero().view(-1)
        keep = nms(boxes[mask], scores[mask], iou_threshold)
        result_mask[mask[keep]] = True
    keep = result_mask.nonzero().view(-1)
    return keep[torch.sort(scores[keep], descending=True, stable=True)[1]]


def nms(boxes: torch.Tensor, scores: torch.Tensor, iou_threshold: float) -> torch.Tensor:
    """torchvision::nms CPU algorithm (oracle_ops.c:oracle_nms)."""
    n = boxes.shape[0]
    if n == 0:
        return torch.empty((0,), dtype=torch.int64)
    order = torch.sort(scores, descending=True, stable=True)[1].contiguous()
    b = boxes.contiguous().float()
    keep = torch.empty(n, dtype=torch.int64)
    k = _lib().oracle_nms(b.data_ptr(), order.data_ptr(), n, float(iou_threshold), keep.data_ptr())
    return keep[:k]


def select_over_all_levels(per_image: dict, nms_thresh=0.6, post_nms_topk=50) -> dict:
    """select_over_all_levels fcos_outputs.py:468-495 + ml_nms layers/ml_nms.py:65-98 for one image:
    boxlist[keep] then topk(scores, min(n, post_nms_topk)).  The kthvalue branch (:485-493) is dead."""
    keep = batched_nms(per_image["boxes"], per_image["scores"], per_image["classes"], nms_thresh)
    res = {k: v[keep] for k, v in per_image.items()}
    n = res["scores"].shape[0]
    k = min(n, post_nms_topk)
    # scores are already in descending (stable) order, so topk == the first k rows; torch.topk's own
    # tie order is unspecified, this fixes it to the stable one.
    idx = torch.arange(k)
    return {key: v[idx] for key, v in res.items()}


def fcos_predict_proposals(logits, bbox_reg, ctrness, strides=(8, 16, 32, 64, 128), pre_nms_thresh=0.05,
                           nms_thresh=0.6, post_nms_topk=50, return_candidates=False, thresh_with_ctr=False):
    """predict_proposals fcos_outputs.py:372-394: per level r*stride (:384), per-level selection,
    Instances.cat per image in level order p3..p7 (:391-392), select_over_all_levels."""
    N = logits[0].shape[0]
    per_level = []
    for l, (o, r, c, s) in enumerate(zip(logits, bbox_reg, ctrness, strides)):
        loc = compute_locations_per_level(o.shape[2], o.shape[3], s)
        per_level.append(fcos_single_level(loc, o, r * s, c, pre_nms_thresh, thresh_with_ctr))
    out, cands = [], []
    for i in range(N):
        cat = {k: torch.cat([lvl[i][k] for lvl in per_level], dim=0) for k in per_level[0][i].keys()}
        cands.append(cat)
        out.append(select_over_all_levels(cat, nms_thresh, post_nms_topk))
    return (out, cands) if return_candidates else out


# --------------------------------------------------------------------------------------
# ROI pooler (pooler.py)
# --------------------------------------------------------------------------------------
def assign_boxes_to_levels_by_ratio(boxes: torch.Tensor, img_areas: torch.Tensor, min_level=3, max_level=5):
    """pooler.py:80-118: ceil(max_level - log2(img_area / box_area + eps)) clamped to [min, max], minus min.
    eps = sys.float_info.epsilon is added to a float32 tensor as in the reference."""
    eps = sys.float_info.epsilon
    box_areas = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
    lv = torch.ceil(max_level - torch.log2(img_areas / box_areas + eps))
    lv = torch.clamp(lv, min=min_level, max=max_level)
    return lv.to(torch.int64) - min_level


def assign_boxes_to_levels(boxes: torch.Tensor, min_level=3, max_level=5, canonical_box_size=224, canonical_level=4):
    """pooler.py:121-152 (ASSIGN_CRITERION "area", FPN Eqn.(1)): floor(canonical_level + log2(sqrt(area) / canonical_box_size + eps))
    clamped to [min, max], minus min."""
    eps = sys.float_info.epsilon
    box_sizes = torch.sqrt((boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1]))
    lv = torch.floor(canonical_level + torch.log2(box_sizes / canonical_box_size + eps))
    lv = torch.clamp(lv, min=min_level, max=max_level)
    return lv.to(torch.int64) - min_level


def roi_align(feat: torch.Tensor, rois: torch.Tensor, scale: float, out_size: int, sampling_ratio: int, aligned: bool):
    """d2 ROIAlign -> torchvision roi_align CPU kernel (oracle_ops.c:oracle_roi_align)."""
    feat = feat.contiguous().float()
    rois = rois.contiguous().float()
    M = rois.shape[0]
    out = torch.zeros((M, feat.shape[1], out_size, out_size), dtype=torch.float32)
    if M:
        _lib().oracle_roi_align(feat.data_ptr(), feat.shape[1], feat.shape[2], feat.shape[3], rois.data_ptr(), M,
                                float(scale), out_size, out_size, int(sampling_ratio), int(bool(aligned)), out.data_ptr())
    return out


def roi_pooler(features: List[torch.Tensor], boxes_per_image: List[torch.Tensor], image_sizes: List[Tuple[int, int]],
               scales=(1 / 8, 1 / 16, 1 / 32), out_size=14, sampling_ratio=0, assign_crit="ratio", aligned=True,
               canonical_box_size=224, canonical_level=4):
    """ROIPooler.forward pooler.py:290-366; pooler_type ROIAlignV2 (aligned) or ROIAlign (:243-255), assign_crit "ratio" (:343)
    or "area".  img_area = image_size[0]*image_size[1] of each Instances (pooler.py:70-77)."""
    rois = torch.cat([torch.cat((torch.full((b.shape[0], 1), float(i)), b), dim=1) for i, b in enumerate(boxes_per_image)], 0)
    img_areas = torch.cat([torch.full((b.shape[0],), float(s[0] * s[1])) for b, s in zip(boxes_per_image, image_sizes)])
    min_level = int(-math.log2(scales[0]))
    max_level = int(-math.log2(scales[-1]))
    if assign_crit == "ratio":
        levels = assign_boxes_to_levels_by_ratio(rois[:, 1:], img_areas, min_level, max_level)
    else:
        levels = assign_boxes_to_levels(rois[:, 1:], min_level, max_level, canonical_box_size, canonical_level)
    out = torch.zeros((rois.shape[0], features[0].shape[1], out_size, out_size), dtype=torch.float32)
    for level, (feat, scale) in enumerate(zip(features, scales)):
        inds = torch.nonzero(levels == level).squeeze(1)
        out[inds] = roi_align(feat, rois[inds], scale, out_size, sampling_ratio, aligned)
    return out, levels


# --------------------------------------------------------------------------------------
# SAG-Mask head (sam.py) and mask selection (mask_head.py)
# --------------------------------------------------------------------------------------
def sam_mask_head_forward(sd, x: torch.Tensor, prefix="roi_heads.mask_head.", num_conv=4) -> torch.Tensor:
    """SpatialAttentionMaskHead.forward sam.py:92-97; SpatialAttention.forward :23-28."""
    for k in range(num_conv):
        x = F.relu(F.conv2d(x, sd[prefix + "mask_fcn{}.weight".format(k + 1)],
                            sd[prefix + "mask_fcn{}.bias".format(k + 1)], padding=1))
    avg_out = torch.mean(x, dim=1, keepdim=True)
    max_out = torch.max(x, dim=1, keepdim=True)[0]
    scale = F.conv2d(torch.cat([avg_out, max_out], dim=1), sd[prefix + "spatialAtt.conv.weight"], None, padding=1)
    x = x * torch.sigmoid(scale)
    x = F.relu(F.conv_transpose2d(x, sd[prefix + "deconv.weight"], sd[prefix + "deconv.bias"], stride=2))
    return F.conv2d(x, sd[prefix + "predictor.weight"], sd[prefix + "predictor.bias"])


def mask_rcnn_inference(mask_logits: torch.Tensor, classes: torch.Tensor) -> torch.Tensor:
    """mask_head.py:197-208 (class-specific branch): sigmoid(logits[arange, cls])[:, None]."""
    idx = torch.arange(mask_logits.shape[0])
    return mask_logits[idx, classes][:, None].sigmoid()


# --------------------------------------------------------------------------------------
# MaskIoU head (maskiou_head.py)
# --------------------------------------------------------------------------------------
def maskiou_head_forward(sd, x: torch.Tensor, mask: torch.Tensor, prefix="roi_heads.maskiou_head.", num_conv=4):
    """MaskIoUHead.forward maskiou_head.py:107-120: maxpool2x2(mask) cat after the 256 ROI channels,
    4 x conv3x3+ReLU (last stride 2), flatten (NCHW order), FC-ReLU, FC-ReLU, FC."""
    mask_pool = F.max_pool2d(mask, kernel_size=2, stride=2)
    x = torch.cat((x, mask_pool), 1)
    for k in range(num_conv):
        stride = 2 if (k + 1) == num_conv else 1
        x = F.relu(F.conv2d(x, sd[prefix + "maskiou_fcn{}.weight".format(k + 1)],
                            sd[prefix + "maskiou_fcn{}.bias".format(k + 1)], stride=stride, padding=1))
    x = torch.flatten(x, 1)
    x = F.relu(F.linear(x, sd[prefix + "maskiou_fc1.weight"], sd[prefix + "maskiou_fc1.bias"]))
    x = F.relu(F.linear(x, sd[prefix + "maskiou_fc2.weight"], sd[prefix + "maskiou_fc2.bias"]))
    return F.linear(x, sd[prefix + "maskiou.weight"], sd[prefix + "maskiou.bias"])


def mask_iou_inference(scores: torch.Tensor, classes: torch.Tensor, pred_maskiou: torch.Tensor) -> torch.Tensor:
    """maskiou_head.py:50-60: mask_scores = scores * maskiou[arange, cls]."""
    return scores * pred_maskiou[torch.arange(pred_maskiou.shape[0]), classes]


# --------------------------------------------------------------------------------------
# End to end (tester.py:25-75 / modified_class.py:28-40 call order)
# --------------------------------------------------------------------------------------
def roi_heads_forward(sd, features: Dict[str, torch.Tensor], proposals: List[dict], image_sizes,
                      in_features=("p3", "p4", "p5"), return_intermediates=False):
    """CenterROIHeads.forward_with_given_boxes center_heads.py:413-444 (mask -> maskiou)."""
    feats = [features[f] for f in in_features]
    boxes = [p["boxes"] for p in proposals]
    roi_feat, levels = roi_pooler(feats, boxes, image_sizes)
    classes = torch.cat([p["classes"] for p in proposals])
    scores = torch.cat([p["scores"] for p in proposals])
    inter = dict(roi_feat=roi_feat, levels=levels)
    if roi_feat.shape[0] == 0:
        pred_masks = torch.zeros((0, 1, 28, 28))
        mask_scores = torch.zeros((0,))
    else:
        mask_logits = sam_mask_head_forward(sd, roi_feat)
        pred_masks = mask_rcnn_inference(mask_logits, classes)
        maskiou = maskiou_head_forward(sd, roi_feat, pred_masks)
        mask_scores = mask_iou_inference(scores, classes, maskiou)
        inter.update(mask_logits=mask_logits, maskiou=maskiou)
    out, start = [], 0
    for p in proposals:
        n = p["boxes"].shape[0]
        q = dict(p)
        q["pred_masks"] = pred_masks[start:start + n]
        q["mask_scores"] = mask_scores[start:start + n]
        out.append(q)
        start += n
    return (out, inter) if return_intermediates else out


def centermask_inference(sd, images: torch.Tensor, image_sizes: List[Tuple[int, int]], conv_body="V-39-eSE",
                         return_intermediates=False):
    """backbone -> proposal_generator -> roi_heads on preprocessed images (N,3,H,W).
    Returns per image dict(boxes, scores, classes, locations, pred_masks, mask_scores), i.e. the fields of
    single_flatten_to_tuple (deploy_utils.py:117-126)."""
    with torch.no_grad():
        feats = backbone_forward(sd, images, conv_body)
        logits, reg, ctr = fcos_head_forward(sd, [feats[k] for k in ("p3", "p4", "p5", "p6", "p7")])
        proposals, cands = fcos_predict_proposals(logits, reg, ctr, return_candidates=True)
        results, inter = roi_heads_forward(sd, feats, proposals, image_sizes, return_intermediates=True)
    if return_intermediates:
        inter.update(features=feats, logits=logits, bbox_reg=reg, ctrness=ctr, candidates=cands, proposals=proposals)
        return results, inter
    return results


def flatten_to_tuple(res: dict):
    """single_flatten_to_tuple deploy_utils.py:117-126 field order."""
    return (res["locations"], res["mask_scores"], res["boxes"], res["classes"], res["pred_masks"], res["scores"])


# --------------------------------------------------------------------------------------
# pre / post-processing (deploy_utils.py:76-98, :129-158) — SURVEY §8(f) rows 1-2
# --------------------------------------------------------------------------------------
def preprocess(image_chw: torch.Tensor, mean=(103.53, 116.28, 123.675), std=(1.0, 1.0, 1.0), fixed_size=1344) -> torch.Tensor:
    """single_preprocessing deploy_utils.py:76-98: (x - mean) / std, zero-pad right/bottom to fixed_size."""
    x = (image_chw.float() - torch.tensor(mean).view(-1, 1, 1)) / torch.tensor(std).view(-1, 1, 1)
    return F.pad(x, (0, fixed_size - x.shape[2], 0, fixed_size - x.shape[1]))


def paste_masks(masks: torch.Tensor, boxes: torch.Tensor, img_h: int, img_w: int, threshold=0.5) -> torch.Tensor:
    """d2 ROIMasks.to_bitmasks -> paste_masks_in_image -> _do_paste_mask (source absent; published algorithm):
    sample the (S,S) mask at every pixel centre with bilinear grid_sample(align_corners=False), then >= threshold."""
    n = masks.shape[0]
    if n == 0:
        return torch.zeros((0, img_h, img_w), dtype=torch.bool)
    x0, y0, x1, y1 = torch.split(boxes, 1, dim=1)
    img_y = torch.arange(0, img_h, dtype=torch.float32) + 0.5
    img_x = torch.arange(0, img_w, dtype=torch.float32) + 0.5
    img_y = (img_y - y0) / (y1 - y0) * 2 - 1
    img_x = (img_x - x0) / (x1 - x0) * 2 - 1
    gx = img_x[:, None, :].expand(n, img_y.size(1), img_x.size(1))
    gy = img_y[:, :, None].expand(n, img_y.size(1), img_x.size(1))
    grid = torch.stack([gx, gy], dim=3)
    img_masks = F.grid_sample(masks[:, None].float(), grid, align_corners=False)
    return img_masks[:, 0] >= threshold


def detector_postprocess(res: dict, h: int, w: int, mask_threshold=0.5) -> dict:
    """deploy_utils.py:129-158."""
    scale = 800 / min(h, w)
    new_h, new_w = int(np.floor(h * scale)), int(np.floor(w * scale))
    if max(new_h, new_w) > 1333:
        scale = 1333 / max(new_h, new_w) * scale
    boxes = res["boxes"].clone()
    boxes[:, 0::2] *= 1 / scale
    boxes[:, 1::2] *= 1 / scale
    boxes = torch.stack((boxes[:, 0].clamp(0, w), boxes[:, 1].clamp(0, h), boxes[:, 2].clamp(0, w), boxes[:, 3].clamp(0, h)), dim=-1)
    keep = ((boxes[:, 2] - boxes[:, 0]) > 0) & ((boxes[:, 3] - boxes[:, 1]) > 0)
    out = {k: v[keep] for k, v in res.items()}
    out["boxes"] = boxes[keep]
    out["pred_masks"] = paste_masks(out["pred_masks"][:, 0], out["boxes"], h, w, mask_threshold)
    return out


def roi_align_boundary_margin(boxes: torch.Tensor, levels: torch.Tensor, feat_hw: List[Tuple[int, int]], scales=(1 / 8, 1 / 16, 1 / 32),
                              out_size: int = 14) -> torch.Tensor:
    """Conditioning of ROIAlign per ROI (test helper): the smallest distance, in feature pixels, of any bilinear sample coordinate to
    the validity boundaries of torchvision's kernel (coordinate == -1 or == size: on one side the sample contributes the clamped
    edge pixel, on the other side 0).  An ROI whose margin is below the box tolerance is discontinuous in the box coordinates for the
    reference op itself; parity tests skip its mask values."""
    out = torch.full((boxes.shape[0],), float("inf"))
    for i in range(boxes.shape[0]):
        lv = int(levels[i])
        h, w = feat_hw[lv]
        s = scales[lv]
        for lo, hi, size in ((float(boxes[i, 0]), float(boxes[i, 2]), w), (float(boxes[i, 1]), float(boxes[i, 3]), h)):
            start, length = lo * s - 0.5, (hi - lo) * s
            grid = max(int(math.ceil(length / out_size)), 0)
            if grid == 0:
                continue
            p = torch.arange(out_size, dtype=torch.float64)[:, None]
            k = torch.arange(grid, dtype=torch.float64)[None, :]
            c = start + p * (length / out_size) + (k + 0.5) * (length / out_size) / grid
            m = torch.minimum((c + 1.0).abs(), (c - size).abs()).min()
            out[i] = min(float(out[i]), float(m))
    return out
