"""FCOS proposal generator on HIP kernels (mirrors centermask2/centermask/modeling/fcos/fcos.py and the inference
half of fcos_outputs.py).

  FCOSHead.forward  fcos.py:222-240   towers = [conv3x3(bias) -> GroupNorm(32) -> ReLU] x4 per branch, on the MFMA conv
                                      kernel + the GN/ReLU kernel; cls_logits as one conv (NHWC output is already the
                                      (N, HW, C) layout the decode wants); bbox_pred and ctrness fused into ONE 256->5
                                      conv whose epilogue applies relu(scale_l * .) to the 4 box channels only.
  predict_proposals fcos_outputs.py:372-495  -> ops.fcos_select (threshold/compact/decode) + ops.nms_topk.
State-dict keys equal the reference's (cls_tower.{0,1,3,4,...}, bbox_tower.*, cls_logits, bbox_pred, ctrness, scales.N.scale).
"""
import math
from typing import Dict, List

import torch
from torch import nn

from ... import ops
from ...ops import View
from ...registry import PROPOSAL_GENERATOR_REGISTRY
from ...structures import Boxes, Instances, ShapeSpec
from ..base import HipModule

__all__ = ["FCOS", "FCOSHead", "Scale"]


class Scale(nn.Module):
    def __init__(self, init_value=1.0):
        super().__init__()
        self.scale = nn.Parameter(torch.FloatTensor([init_value]))


class FCOSHead(HipModule):
    def __init__(self, cfg, input_shape: List[ShapeSpec]):
        super().__init__()
        self.num_classes = cfg.MODEL.FCOS.NUM_CLASSES
        self.fpn_strides = cfg.MODEL.FCOS.FPN_STRIDES
        if cfg.MODEL.FCOS.USE_DEFORMABLE:
            raise NotImplementedError("MODEL.FCOS.USE_DEFORMABLE (off in the reference config, defaults.py:41)")
        self.norm = None if cfg.MODEL.FCOS.NORM == "none" else cfg.MODEL.FCOS.NORM
        if self.norm not in (None, "GN"):
            raise NotImplementedError("MODEL.FCOS.NORM={}".format(self.norm))
        head_configs = {"cls": cfg.MODEL.FCOS.NUM_CLS_CONVS, "bbox": cfg.MODEL.FCOS.NUM_BOX_CONVS, "share": cfg.MODEL.FCOS.NUM_SHARE_CONVS}
        in_channels = [s.channels for s in input_shape]
        assert len(set(in_channels)) == 1, "Each level must have the same channel!"
        in_channels = in_channels[0]
        self.in_channels = in_channels
        for head, num_convs in head_configs.items():
            tower = []
            for _ in range(num_convs):
                tower.append(nn.Conv2d(in_channels, in_channels, kernel_size=3, stride=1, padding=1, bias=True))
                if self.norm == "GN":
                    tower.append(nn.GroupNorm(32, in_channels))
                tower.append(nn.ReLU())
            self.add_module("{}_tower".format(head), nn.Sequential(*tower))
        self.cls_logits = nn.Conv2d(in_channels, self.num_classes, kernel_size=3, stride=1, padding=1)
        self.bbox_pred = nn.Conv2d(in_channels, 4, kernel_size=3, stride=1, padding=1)
        self.ctrness = nn.Conv2d(in_channels, 1, kernel_size=3, stride=1, padding=1)
        self.scales = nn.ModuleList([Scale(init_value=1.0) for _ in self.fpn_strides]) if cfg.MODEL.FCOS.USE_SCALE else None
        for modules in [self.cls_tower, self.bbox_tower, self.share_tower, self.cls_logits, self.bbox_pred, self.ctrness]:
            for l in modules.modules():
                if isinstance(l, nn.Conv2d):
                    torch.nn.init.normal_(l.weight, std=0.01)
                    torch.nn.init.constant_(l.bias, 0)
        prior_prob = cfg.MODEL.FCOS.PRIOR_PROB
        torch.nn.init.constant_(self.cls_logits.bias, -math.log((1 - prior_prob) / prior_prob))

    def _tower_packed(self, tower: nn.Sequential, dev):
        out = []
        mods = list(tower)
        i = 0
        while i < len(mods):
            conv = mods[i]
            assert isinstance(conv, nn.Conv2d)
            gn = mods[i + 1] if (i + 1 < len(mods) and isinstance(mods[i + 1], nn.GroupNorm)) else None
            pc = ops.PackedConv(conv.weight, None, conv.bias, dev)
            if gn is not None:
                out.append((pc, gn.weight.detach().float().contiguous().to(dev), gn.bias.detach().float().contiguous().to(dev), gn.eps, gn.num_groups))
                i += 3
            else:
                out.append((pc, None, None, 0.0, 0))
                i += 2
        return out

    def _build_packed(self, dev):
        P = {"share": self._tower_packed(self.share_tower, dev), "cls": self._tower_packed(self.cls_tower, dev),
             "bbox": self._tower_packed(self.bbox_tower, dev)}
        P["cls_logits"] = ops.PackedConv(self.cls_logits.weight, None, self.cls_logits.bias, dev)
        # bbox_pred (4) + ctrness (1) share one conv; per level: (acc + b) * s = acc * s + b * s on the box channels
        w = torch.cat([self.bbox_pred.weight.detach().float().cpu(), self.ctrness.weight.detach().float().cpu()], 0)
        b = torch.cat([self.bbox_pred.bias.detach().float().cpu(), self.ctrness.bias.detach().float().cpu()], 0)
        base = ops.PackedConv(w, None, None, dev)
        P["regctr"] = []
        for l in range(len(self.fpn_strides)):
            s = float(self.scales[l].scale.detach().float().cpu()) if self.scales is not None else 1.0
            pc = ops.PackedConv.__new__(ops.PackedConv)
            pc.__dict__.update(base.__dict__)
            sc = torch.tensor([s, s, s, s, 1.0], dtype=torch.float32)
            pc.scale = sc.to(dev)
            pc.shift = (b * sc).to(dev)
            P["regctr"].append(pc)
        return P

    @staticmethod
    def _run_tower(xs: List[View], tower, aff=None):
        """One launch per tower conv over ALL levels (the weights are shared, fcos.py:227-231).  GroupNorm+ReLU is not a pass
        of its own: its statistics become a per-(image, channel) affine that the NEXT conv applies while staging its input.
        Returns (raw conv outputs, pending affine or None)."""
        for pc, gamma, beta, eps, groups in tower:
            if gamma is None:
                xs = ops.conv_out_multi(xs, [pc] * len(xs), relu=True, in_affine=aff)
                aff = None
            else:
                xs, aff = ops.conv_gn_multi(xs, [pc] * len(xs), gamma, beta, groups, eps, in_affine=aff)
        return xs, aff

    @classmethod
    def _run_tower_pair(cls, xs: List[View], tower_a, tower_b, aff=None):
        """The cls and the bbox tower side by side (fcos.py:227-231: same input, same shapes, different weights): conv k of both towers
        is ONE launch of 2 x levels problems where the F(4x4) kernel runs it (ops.conv_gn_multi_pair) — half the launch ramps and tails;
        otherwise (other kernels, towers of different depth, no GroupNorm) the towers run one after the other as before."""
        same = len(tower_a) == len(tower_b) and all(ta[1] is not None and tb[1] is not None and ta[3] == tb[3] and ta[4] == tb[4]
                                                    for ta, tb in zip(tower_a, tower_b))
        if not same:
            a, aa = cls._run_tower(xs, tower_a, aff)
            b, ba = cls._run_tower(xs, tower_b, aff)
            return a, aa, b, ba
        xa, xb, aa, ab = xs, xs, aff, aff
        for (pca, ga, bea, eps, groups), (pcb, gb, beb, _, _) in zip(tower_a, tower_b):
            pair = ops.conv_gn_multi_pair(xa, pca, (ga, bea), xb, pcb, (gb, beb), groups, eps, in_affine_a=aa, in_affine_b=ab)
            if pair is None:
                xa, aa = ops.conv_gn_multi(xa, [pca] * len(xa), ga, bea, groups, eps, in_affine=aa)
                xb, ab = ops.conv_gn_multi(xb, [pcb] * len(xb), gb, beb, groups, eps, in_affine=ab)
            else:
                (xa, aa), (xb, ab) = pair
        return xa, aa, xb, ab

    def forward_views(self, feats: List[View]):
        """-> (logits[l] (N,H,W,C) NHWC, regctr[l] (N,H,W,5) = [relu(scale_l*bbox_pred) x4, ctrness logit])."""
        P = self.packed()
        nl = len(feats)
        out_l, out_r = [], []
        for g0 in range(0, nl, 5):                      # the multi-problem launch takes up to 5 levels
            f, fa = self._run_tower(list(feats[g0:g0 + 5]), P["share"])
            cls_t, ca, box_t, ba = self._run_tower_pair(f, P["cls"], P["bbox"], fa)
            out_l += [v.t for v in ops.conv_out_multi(cls_t, [P["cls_logits"]] * len(cls_t), in_affine=ca)]
            out_r += [v.t for v in ops.conv_out_multi(box_t, P["regctr"][g0:g0 + 5], relu_upto=4, in_affine=ba)]       # fcos.py:233-238
        return out_l, out_r

    def forward(self, x: List[torch.Tensor]):
        """Reference signature (fcos.py:222-240): NCHW logits, bbox_reg, ctrness, bbox_towers([])."""
        logits, regctr = self.forward_views([ops.as_view(f) for f in x])
        lg = [t.permute(0, 3, 1, 2) for t in logits]
        reg = [t[..., :4].permute(0, 3, 1, 2) for t in regctr]
        ctr = [t[..., 4:5].permute(0, 3, 1, 2) for t in regctr]
        return lg, reg, ctr, []


@PROPOSAL_GENERATOR_REGISTRY.register()
class FCOS(HipModule):
    def __init__(self, cfg, input_shape: Dict[str, ShapeSpec]):
        super().__init__()
        self.in_features = cfg.MODEL.FCOS.IN_FEATURES
        self.fpn_strides = cfg.MODEL.FCOS.FPN_STRIDES
        self.pre_nms_thresh_test = cfg.MODEL.FCOS.INFERENCE_TH_TEST
        self.pre_nms_topk_test = cfg.MODEL.FCOS.PRE_NMS_TOPK_TEST      # unused by the fork (fcos_outputs.py:444-449)
        self.nms_thresh = cfg.MODEL.FCOS.NMS_TH
        self.post_nms_topk_test = cfg.MODEL.FCOS.POST_NMS_TOPK_TEST
        self.thresh_with_ctr = cfg.MODEL.FCOS.THRESH_WITH_CTR          # fcos_outputs.py:410-420 (False in the reference recipe)
        if not (1 <= self.post_nms_topk_test <= 1024):
            raise NotImplementedError("POST_NMS_TOPK_TEST must be in [1, 1024] (the NMS kernel's keep list lives in LDS)")
        self.mask_on = cfg.MODEL.MASK_ON
        self.fcos_head = FCOSHead(cfg, [input_shape[f] for f in self.in_features])
        # candidates per image the workspaces are sized for.  The reference is unbounded (no pre-NMS top-k, fcos_outputs.py:444-449):
        # the select kernel reports the TRUE count, `overflow` rides along with the padded results, and wherever the counts are read
        # on the host (forward(), results_from_padded) an overflow re-runs select + NMS with a capacity sized from the true count
        # and keeps the larger capacity for later calls — never a silent truncation, never an error.
        self.candidate_capacity = 131072

    def _build_packed(self, dev):
        return {}

    def compute_locations(self, features):
        """fcos.py:120-144 (kept for API parity; the decode kernel derives locations from the index)."""
        locations = []
        for level, feature in enumerate(features):
            h, w = feature.size()[-2:]
            s = self.fpn_strides[level]
            shifts_x = torch.arange(0, w * s, step=s, dtype=torch.float32, device=feature.device)
            shifts_y = torch.arange(0, h * s, step=s, dtype=torch.float32, device=feature.device)
            shift_y, shift_x = torch.meshgrid(shifts_y, shifts_x, indexing="ij")
            locations.append(torch.stack((shift_x.reshape(-1), shift_y.reshape(-1)), dim=1) + s // 2)
        return locations

    def _select_nms(self, logits, regctr, cap: int):
        cand = ops.fcos_select(logits, regctr, self.fpn_strides, self.pre_nms_thresh_test, cap, self.thresh_with_ctr)
        det = ops.nms_topk(cand, self.nms_thresh, self.post_nms_topk_test)
        det["cand_counts"] = cand["counts"]
        det["cand_capacity"] = cap
        det["overflow"] = cand["counts"] > cap                       # (N,) bool on the device: candidates beyond `cap` were dropped
        det["_rerun"] = lambda new_cap: self._select_nms(logits, regctr, new_cap)
        return det, cand

    def forward_padded(self, features):
        """Device-only path: returns (detections dict padded to (N, topk), candidates dict).  No host sync.  det["overflow"] tells
        (on the device) whether an image had more candidates than the capacity; resolve_overflow() acts on it at the caller's sync."""
        feats = [ops.as_view(features[f]) for f in self.in_features]
        logits, regctr = self.fcos_head.forward_views(feats)
        return self._select_nms(logits, regctr, self.candidate_capacity)

    def resolve_overflow(self, det: dict) -> dict:
        """One host sync on the candidate counts: if any image overflowed, select + NMS run again with a capacity sized from the
        true count (the head is not recomputed) and the capacity is kept for later calls."""
        worst = int(det["cand_counts"].max())
        if worst <= det["cand_capacity"]:
            return det
        new_cap = 1 << (worst - 1).bit_length()
        self.candidate_capacity = max(self.candidate_capacity, new_cap)
        det2, _ = det["_rerun"](new_cap)
        assert int(det2["cand_counts"].max()) <= new_cap
        return det2

    def forward(self, images, features, gt_instances=None):
        """fcos.py:61-118 (inference branch): -> (list[Instances], {}).  `images` needs len() and .image_sizes.
        Nothing is read back here: the Instances are LAZY (LazyInstances below) — their fields are sliced out of the padded device
        buffers on first access, which is also where the candidate-overflow check happens — so a caller that hands them straight to
        CenterROIHeads (tester.py:57-66, modified_class.py:35-38) keeps the GPU fed across the two plugins."""
        if self.training:
            raise NotImplementedError("training is out of scope of the MI355X inference path")
        det, _ = self.forward_padded(features)
        batch = LazyBatch(det, images.image_sizes, self)
        return [LazyInstances(tuple(hw), batch, i) for i, hw in enumerate(images.image_sizes)], {}


class LazyBatch:
    """The padded device buffers behind a list of LazyInstances (one FCOS call) and, once CenterROIHeads has run on them, the ROI
    outputs.  resolve() is the single host synchronisation: it reads the candidate counts, re-runs select + NMS (+ the ROI heads)
    with a larger capacity if an image overflowed — the reference is unbounded (fcos_outputs.py:444-449) — and reads the per-image
    detection counts."""

    def __init__(self, det: dict, image_sizes, fcos):
        self.det, self.image_sizes, self.fcos = det, [tuple(hw) for hw in image_sizes], fcos
        self.out = None          # ROI-head outputs over the same padded buffers (pred_masks, mask_scores), if any
        self.redo = None         # det -> ROI-head outputs, for the overflow re-run
        self.counts = None       # per-image detection counts on the host, once resolved

    def attach_roi(self, out: dict, redo) -> None:
        self.out, self.redo = out, redo

    def resolve(self) -> None:
        if self.counts is not None:
            return
        det = self.det
        if "cand_capacity" in det and int(det["cand_counts"].max()) > det["cand_capacity"]:
            det = self.fcos.resolve_overflow(det)
            self.det = det
            if self.out is not None:
                self.out = self.redo(det)
        self.counts = det["counts"].cpu().tolist()

    def fields_of(self, i: int) -> dict:
        self.resolve()
        det, k = self.det, self.counts[i]
        f = {"pred_boxes": Boxes(det["box"][i, :k]), "scores": det["score"][i, :k], "pred_classes": det["cls"][i, :k],
             "locations": det["loc"][i, :k]}
        if self.out is not None:
            for name in ("pred_masks", "mask_scores"):
                if name in self.out:
                    f[name] = self.out[name][i, :k]
        return f


class LazyInstances(Instances):
    """Instances (fields of fcos_outputs.py:458-462) whose fields are sliced out of the batch's padded device buffers on first access.
    Behaves like Instances in every other respect; indexing / .to() return plain Instances."""

    def __init__(self, image_size, batch: LazyBatch, index: int):
        object.__setattr__(self, "_image_size", image_size)
        object.__setattr__(self, "_cmk_lazy", (batch, index))
        object.__setattr__(self, "_store", None)

    @property
    def _fields(self):
        st = self.__dict__.get("_store")
        if st is None:
            batch, i = self.__dict__.get("_cmk_lazy", (None, -1))       # absent while copy / pickle rebuild the object: no fields yet
            st = batch.fields_of(i) if batch is not None else {}
            object.__setattr__(self, "_store", st)
        return st

    def set(self, name: str, value) -> None:
        if name in ("pred_boxes", "pred_classes", "scores", "locations"):
            _ = self._fields                                     # materialise first, then the caller's value replaces ours ...
            object.__setattr__(self, "_cmk_lazy", (None, -1))    # ... and the padded buffers no longer describe this object
        Instances.set(self, name, value)

    def lazy_batch(self):
        """(LazyBatch, index) while the padded buffers still describe this object, else (None, -1)."""
        return self.__dict__.get("_cmk_lazy", (None, -1))


def instances_from_padded(det: dict, image_sizes) -> List[Instances]:
    """One host sync: read the per-image counts, slice the padded device buffers into Instances
    (fields of fcos_outputs.py:458-462).  Overflow must have been resolved (FCOS.resolve_overflow) — a truncated candidate set is
    refused here rather than handed on."""
    counts = det["counts"].cpu().tolist()
    if "cand_capacity" in det and int(det["cand_counts"].max()) > det["cand_capacity"]:
        raise RuntimeError("FCOS candidates overflowed the capacity {}: call FCOS.resolve_overflow(det) first".format(det["cand_capacity"]))
    out = []
    for i, k in enumerate(counts):
        inst = Instances(tuple(image_sizes[i]))
        inst.pred_boxes = Boxes(det["box"][i, :k])
        inst.scores = det["score"][i, :k]
        inst.pred_classes = det["cls"][i, :k]
        inst.locations = det["loc"][i, :k]
        out.append(inst)
    return out
