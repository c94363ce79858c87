"""Output post-processing of the reference's deployment path (deploy_utils.py:129-175), device side.

detector_postprocess: rescale boxes back to the original resolution (the inverse of the 800/1333 shortest-edge resize),
clip to the image, drop empty boxes, paste the 28x28 masks into full-image bitmasks at 0.5.
"""
import numpy as np
import torch

from . import ops
from .structures import Boxes, Instances

MIN_EDGE_SIZE = 800       # deploy_utils.py:19-21
MAX_EDGE_SIZE = 1333
FIXED_EDGE_SIZE = 1344


def resize_scale(h: int, w: int) -> float:
    """The scale the reference's loader applied (ResizeShortestEdge(800, 1333)), deploy_utils.py:138-142."""
    scale = MIN_EDGE_SIZE / min(h, w)
    new_h = int(np.floor(h * scale))
    new_w = int(np.floor(w * scale))
    if max(new_h, new_w) > MAX_EDGE_SIZE:
        scale = MAX_EDGE_SIZE / max(new_h, new_w) * scale
    return scale


def detector_postprocess(results: Instances, h: int, w: int, mask_threshold: float = 0.5) -> Instances:
    results = Instances((h, w), **results.get_fields())
    scale = resize_scale(h, w)
    scale_x, scale_y = 1 / scale, 1 / scale
    output_boxes = results.pred_boxes.clone()
    output_boxes.scale(scale_x, scale_y)
    output_boxes.clip(results.image_size)
    results.pred_boxes = output_boxes
    results = results[output_boxes.nonempty()]
    if results.has("pred_masks"):
        results.pred_masks = ops.paste_masks(results.pred_masks[:, 0, :, :], results.pred_boxes.tensor, h, w, mask_threshold)
    return results


def postprocess(instances, height=MAX_EDGE_SIZE, width=MAX_EDGE_SIZE):
    """deploy_utils.py:161-175."""
    return [{"instances": detector_postprocess(r, height, width)} for r in instances]
