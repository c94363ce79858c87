"""Output post-processing, device side.  Two callers in the reference, two scale rules:

  detector_postprocess     deploy_utils.py:129-158 (the ONNX/.bin deployment path): the scale is RECOMPUTED from the output size as the
                           inverse of ResizeShortestEdge(800, 1333) — only right for inputs that went through that resize;
  detector_postprocess_d2  detectron2's detector_postprocess, which GeneralizedRCNN._postprocess calls (tester.py:73): boxes scale by
                           (out_w / image_size[1], out_h / image_size[0]) of the Instances they came with.
Both then clip to the image, drop empty boxes and paste the 28x28 masks into full-image bitmasks at 0.5 (cmk_paste_masks).
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


def detector_postprocess_d2(results: Instances, output_height: int, output_width: int, mask_threshold: float = 0.5) -> Instances:
    """detectron2.modeling.postprocessing.detector_postprocess (source absent; published behaviour), as reached from
    GeneralizedRCNN._postprocess (tester.py:73): results.image_size is the size the network saw."""
    scale_x, scale_y = output_width / results.image_size[1], output_height / results.image_size[0]
    results = Instances((output_height, output_width), **results.get_fields())
    output_boxes = results.pred_boxes.clone()
    output_boxes.scale(scale_x, scale_y)
    output_boxes.clip(results.image_size)
    results.pred_boxes = output_boxes
    results = results[output_boxes.nonempty()]
    if results.has("pred_masks"):
        results.pred_masks = ops.paste_masks(results.pred_masks[:, 0, :, :], results.pred_boxes.tensor, output_height, output_width, mask_threshold)
    return results


def postprocess(instances, height=MAX_EDGE_SIZE, width=MAX_EDGE_SIZE):
    """deploy_utils.py:161-175."""
    return [{"instances": detector_postprocess(r, height, width)} for r in instances]
