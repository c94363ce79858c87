"""Result wire formats of the reference's tooling, host side, no third-party dependency.

* the six `.bin` files per image of the Ascend flow (postprocess_bin_outputs.py:36-45): `<name>_1.bin` .. `_6.bin` =
  locations f32 (-1,2), mask_scores f32 (-1), pred_boxes f32 (-1,4), pred_classes i64 (-1), pred_masks f32 (-1,1,28,28),
  scores f32 (-1) — the order of single_flatten_to_tuple (deploy_utils.py:117-126);
* COCO result json (evaluation/coco_evaluation.py:362-427): XYWH boxes, compressed RLE segmentation, `mask_score`, and the rule
  that segm AP is ranked by `mask_score` (:557-563).
pycocotools is absent here, so the RLE codec below restates its published algorithm (maskApi.c rleEncode / rleToString:
column-major runs starting with a zero run, 5-bit groups with a continuation bit, delta coding from the third count on);
it is checked by round trip only ("parity unpinned" for the exact bytes).
"""
import os
from typing import Dict, List, Tuple

import numpy as np
import torch

BIN_DTYPES = ("float32", "float32", "float32", "int64", "float32", "float32")
BIN_SHAPES = ((-1, 2), (-1,), (-1, 4), (-1,), (-1, 1, 28, 28), (-1,))


def to_bin(tuple6, prefix: str) -> List[str]:
    """tuple6 = (locations, mask_scores, pred_boxes, pred_classes, pred_masks, scores); writes prefix_1.bin .. prefix_6.bin."""
    paths = []
    for i, (t, dt) in enumerate(zip(tuple6, BIN_DTYPES)):
        path = "{}_{}.bin".format(prefix, i + 1)
        np.ascontiguousarray(t.detach().cpu().numpy().astype(dt)).tofile(path)
        paths.append(path)
    return paths


def from_bin(prefix: str) -> tuple:
    out = []
    for i, (dt, shp) in enumerate(zip(BIN_DTYPES, BIN_SHAPES)):
        out.append(torch.from_numpy(np.fromfile("{}_{}.bin".format(prefix, i + 1), dtype=dt).reshape(shp)))
    return tuple(out)


def rle_counts(mask: np.ndarray) -> List[int]:
    """Run lengths of the column-major flattening, starting with the zeros run (possibly 0 long)."""
    flat = np.asarray(mask, dtype=np.uint8).reshape(-1, order="F")
    if flat.size == 0:
        return []
    change = np.flatnonzero(flat[1:] != flat[:-1]) + 1
    bounds = np.concatenate(([0], change, [flat.size]))
    counts = np.diff(bounds).tolist()
    if flat[0] == 1:
        counts = [0] + counts
    return counts


def rle_to_string(counts: List[int]) -> str:
    s = []
    for i, x in enumerate(counts):
        x = int(x)
        if i > 2:
            x -= int(counts[i - 2])
        more = True
        while more:
            c = x & 0x1F
            x >>= 5
            more = (x != -1) if (c & 0x10) else (x != 0)
            if more:
                c |= 0x20
            s.append(chr(c + 48))
    return "".join(s)


def rle_from_string(s: str) -> List[int]:
    counts, p = [], 0
    while p < len(s):
        x, k, more = 0, 0, True
        while more:
            c = ord(s[p]) - 48
            x |= (c & 0x1F) << (5 * k)
            more = bool(c & 0x20)
            p += 1
            k += 1
            if not more and (c & 0x10):
                x |= -1 << (5 * k)
        if len(counts) > 2:
            x += counts[-2]
        counts.append(x)
    return counts


def rle_encode(mask) -> Dict:
    m = mask.detach().cpu().numpy() if torch.is_tensor(mask) else np.asarray(mask)
    return {"size": [int(m.shape[0]), int(m.shape[1])], "counts": rle_to_string(rle_counts(m))}


def rle_decode(rle: Dict) -> np.ndarray:
    h, w = rle["size"]
    counts = rle_from_string(rle["counts"])
    flat = np.zeros(h * w, dtype=np.uint8)
    pos, val = 0, 0
    for c in counts:
        flat[pos:pos + c] = val
        pos += c
        val ^= 1
    return flat.reshape((h, w), order="F").astype(bool)


def instances_to_coco_json(instances, img_id: int) -> List[Dict]:
    """coco_evaluation.py:362-427 for post-processed Instances (pred_masks = (n,H,W) bool bitmasks)."""
    n = len(instances)
    if n == 0:
        return []
    boxes = instances.pred_boxes.tensor.detach().cpu().clone()
    boxes[:, 2] -= boxes[:, 0]          # XYXY_ABS -> XYWH_ABS
    boxes[:, 3] -= boxes[:, 1]
    boxes = boxes.tolist()
    scores = instances.scores.tolist()
    classes = instances.pred_classes.tolist()
    rles = [rle_encode(m) for m in instances.pred_masks] if instances.has("pred_masks") else None
    mask_scores = instances.mask_scores.tolist() if instances.has("mask_scores") else None
    results = []
    for k in range(n):
        r = {"image_id": img_id, "category_id": classes[k], "bbox": boxes[k], "score": scores[k]}
        if rles is not None:
            r["segmentation"] = rles[k]
            if mask_scores is not None:
                r["mask_score"] = mask_scores[k]
        results.append(r)
    return results


def segm_results_ranked_by_mask_score(coco_results: List[Dict]) -> List[Dict]:
    """coco_evaluation.py:551-563: for segm AP drop `bbox` and let `mask_score` replace `score`."""
    out = []
    for c in coco_results:
        c = dict(c)
        c.pop("bbox", None)
        if "mask_score" in c:
            c["score"] = c.pop("mask_score")
        out.append(c)
    return out
