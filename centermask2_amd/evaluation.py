"""Self-contained COCO-style AP for fixed inputs (the "AP delta" half of the metric).

COCO val2017 and pycocotools are not available offline, so AP is computed against a pseudo ground truth (the detections of
the CPU restatement of the reference on the same images): AP@[.50:.05:.95], 101-point interpolation, per class then
averaged, boxes by box IoU and masks by the IoU of the pasted bitmasks.  As in the reference's evaluator
(evaluation/coco_evaluation.py:557-563) segmentation predictions are ranked by `mask_scores`, boxes by `scores`.
"""
from typing import Dict, List

import numpy as np
import torch


def box_iou(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    area_a = (a[:, 2] - a[:, 0]).clamp(min=0) * (a[:, 3] - a[:, 1]).clamp(min=0)
    area_b = (b[:, 2] - b[:, 0]).clamp(min=0) * (b[:, 3] - b[:, 1]).clamp(min=0)
    lt = torch.max(a[:, None, :2], b[None, :, :2])
    rb = torch.min(a[:, None, 2:], b[None, :, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    return inter / (area_a[:, None] + area_b[None, :] - inter).clamp(min=1e-12)


def mask_iou(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """a (n,H,W) bool, b (m,H,W) bool."""
    af, bf = a.flatten(1).float(), b.flatten(1).float()
    inter = af @ bf.t()
    union = af.sum(1)[:, None] + bf.sum(1)[None, :] - inter
    iou = inter / union.clamp(min=1.0)
    return torch.where(union == 0, torch.ones_like(iou), iou)      # two empty masks (box outside the image) coincide


def average_precision(preds: List[Dict[str, torch.Tensor]], gts: List[Dict[str, torch.Tensor]], kind: str = "box") -> float:
    """preds[i] / gts[i]: per image dict(boxes (n,4), classes (n), scores (n) [, masks (n,H,W) bool, mask_scores (n)])."""
    thresholds = np.arange(0.5, 0.96, 0.05)
    classes = sorted(set(int(c) for g in gts for c in g["classes"].tolist()))
    aps = []
    for c in classes:
        per_thr = []
        for thr in thresholds:
            scores, tps, n_gt = [], [], 0
            for p, g in zip(preds, gts):
                gi = (g["classes"] == c).nonzero().flatten()
                pi = (p["classes"] == c).nonzero().flatten()
                n_gt += len(gi)
                if len(pi) == 0:
                    continue
                sc = (p["mask_scores"] if kind == "mask" and "mask_scores" in p else p["scores"])[pi]
                order = torch.argsort(sc, descending=True, stable=True)
                pi, sc = pi[order], sc[order]
                if len(gi):
                    iou = box_iou(p["boxes"][pi], g["boxes"][gi]) if kind == "box" else mask_iou(p["masks"][pi], g["masks"][gi])
                taken = torch.zeros(len(gi), dtype=torch.bool)
                for k in range(len(pi)):
                    hit = False
                    if len(gi):
                        cand = iou[k].clone()
                        cand[taken] = -1
                        j = int(torch.argmax(cand))
                        if cand[j] >= thr - 1e-9:
                            taken[j] = True
                            hit = True
                    scores.append(float(sc[k]))
                    tps.append(hit)
            if n_gt == 0:
                continue
            if not scores:
                per_thr.append(0.0)
                continue
            o = np.argsort(-np.asarray(scores), kind="stable")
            tp = np.cumsum(np.asarray(tps, dtype=np.float64)[o])
            fp = np.cumsum(1.0 - np.asarray(tps, dtype=np.float64)[o])
            recall = tp / n_gt
            precision = tp / np.maximum(tp + fp, 1e-12)
            for i in range(len(precision) - 2, -1, -1):
                precision[i] = max(precision[i], precision[i + 1])
            rs = np.linspace(0, 1, 101)
            idx = np.searchsorted(recall, rs, side="left")
            per_thr.append(float(np.mean([precision[i] if i < len(precision) else 0.0 for i in idx])))
        if per_thr:
            aps.append(float(np.mean(per_thr)))
    return float(np.mean(aps)) if aps else float("nan")
