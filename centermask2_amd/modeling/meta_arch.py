"""GeneralizedRCNN + build_model(cfg): the detectron2 meta-architecture surface the reference drives
(tester.py:24-75,157-167; modified_class.py:27-40; deploy_utils.py:76-126).

The three plugins are resolved by name from the registries exactly as detectron2's build_model does:
cfg.MODEL.BACKBONE.NAME -> BACKBONE_REGISTRY, cfg.MODEL.PROPOSAL_GENERATOR.NAME -> PROPOSAL_GENERATOR_REGISTRY,
cfg.MODEL.ROI_HEADS.NAME -> ROI_HEADS_REGISTRY.  The fast path (`inference_padded`) keeps every count on the device and
syncs once when the results are handed over.
"""
from typing import Dict, List, Optional, Sequence, Tuple

import torch
from torch import nn

from ..registry import BACKBONE_REGISTRY, META_ARCH_REGISTRY, PROPOSAL_GENERATOR_REGISTRY, ROI_HEADS_REGISTRY
from ..structures import Boxes, FakeImageList, ImageList, Instances, ShapeSpec
from .fcos.fcos import instances_from_padded

__all__ = ["GeneralizedRCNN", "build_model", "build_backbone", "build_proposal_generator", "build_roi_heads", "flatten_to_tuple"]


def build_backbone(cfg, input_shape=None):
    if input_shape is None:
        input_shape = ShapeSpec(channels=len(cfg.MODEL.PIXEL_MEAN))
    return BACKBONE_REGISTRY.get(cfg.MODEL.BACKBONE.NAME)(cfg, input_shape)


def build_proposal_generator(cfg, input_shape):
    name = cfg.MODEL.PROPOSAL_GENERATOR.NAME
    if name == "PrecomputedProposals":
        return None
    return PROPOSAL_GENERATOR_REGISTRY.get(name)(cfg, input_shape)


def build_roi_heads(cfg, input_shape):
    return ROI_HEADS_REGISTRY.get(cfg.MODEL.ROI_HEADS.NAME)(cfg, input_shape)


def flatten_to_tuple(inst: Instances) -> tuple:
    """single_flatten_to_tuple deploy_utils.py:117-126: (locations, mask_scores, pred_boxes, pred_classes, pred_masks, scores)."""
    f = inst.get_fields()
    return (f["locations"], f["mask_scores"], f["pred_boxes"].tensor, f["pred_classes"], f["pred_masks"], f["scores"])


class GeneralizedRCNN(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.backbone = build_backbone(cfg)
        self.proposal_generator = build_proposal_generator(cfg, self.backbone.output_shape())
        self.roi_heads = build_roi_heads(cfg, self.backbone.output_shape())
        self.input_format = cfg.INPUT.FORMAT
        self.register_buffer("pixel_mean", torch.tensor(cfg.MODEL.PIXEL_MEAN).view(-1, 1, 1), False)
        self.register_buffer("pixel_std", torch.tensor(cfg.MODEL.PIXEL_STD).view(-1, 1, 1), False)

    @property
    def device(self):
        return self.pixel_mean.device

    def preprocess_image(self, batched_inputs: List[Dict[str, torch.Tensor]]) -> ImageList:
        """Normalise and batch: (x - mean) / std, zero-pad right/bottom to the backbone's size divisibility
        (d2 GeneralizedRCNN.preprocess_image; deploy_utils.py:76-98 pads to a fixed 1344 instead)."""
        from .. import ops
        images = [x["image"].to(self.device) for x in batched_inputs]
        images = [im if im.dtype in (torch.uint8, torch.float32) else im.float() for im in images]
        batch, sizes = ops.preprocess_images(images, self.pixel_mean.flatten().tolist(), self.pixel_std.flatten().tolist(),
                                             self.backbone.size_divisibility)
        return ImageList(batch, sizes)

    # -- device-only fast path --------------------------------------------------------------------------------------
    def inference_padded(self, images: torch.Tensor, image_sizes: Sequence[Tuple[int, int]], want=()) -> dict:
        """images: preprocessed (N,3,H,W) on the GPU.  Returns padded device buffers (box, score, cls, loc, counts,
        pred_masks, mask_scores, cand_counts, overflow); no host synchronisation happens here.  `overflow` (N bools on the device)
        says whether an image had more FCOS candidates than the capacity the buffers were sized for: results_from_padded()
        — the caller's sync point — then re-runs the detection tail and the ROI heads with a larger capacity."""
        features = self.backbone(images)
        det, _ = self.proposal_generator.forward_padded(features)
        out = self.roi_heads.forward_padded(features, det, image_sizes, want=want)
        out["_redo"] = lambda det2: self.roi_heads.forward_padded(features, det2, image_sizes, want=want)
        return out

    def results_from_padded(self, out: dict, image_sizes) -> List[Instances]:
        if "cand_capacity" in out and int(out["cand_counts"].max()) > out["cand_capacity"]:      # unbounded like the reference
            out = out["_redo"](self.proposal_generator.resolve_overflow(out))
        insts = instances_from_padded(out, image_sizes)
        for i, it in enumerate(insts):
            m = len(it)
            if "pred_masks" in out:
                it.pred_masks = out["pred_masks"][i, :m]
            if "mask_scores" in out:
                it.mask_scores = out["mask_scores"][i, :m]
        return insts

    # -- reference API ----------------------------------------------------------------------------------------------
    def inference(self, batched_inputs, detected_instances=None, do_preprocess: bool = True, do_postprocess: bool = True):
        """tester.py:25-75.  With do_preprocess=False `batched_inputs` is an ImageList/FakeImageList."""
        assert not self.training
        images = self.preprocess_image(batched_inputs) if do_preprocess else batched_inputs
        features = self.backbone(images.tensor)
        if detected_instances is None:
            proposals, _ = self.proposal_generator(images, features, None)
            results, _ = self.roi_heads(images, features, proposals, None)
        else:
            detected_instances = [x.to(self.device) for x in detected_instances]
            results = self.roi_heads.forward_with_given_boxes(features, detected_instances)
        if do_postprocess:      # GeneralizedRCNN._postprocess (tester.py:73): d2's detector_postprocess, scale = output size / network input size
            from ..postprocess import detector_postprocess_d2
            out = []
            for r, inp, size in zip(results, batched_inputs, images.image_sizes):
                out.append({"instances": detector_postprocess_d2(r, inp.get("height", size[0]), inp.get("width", size[1]))})
            return out
        return results

    def forward(self, batched_inputs):
        if torch.is_tensor(batched_inputs):
            return self.forward_tensor(batched_inputs)
        return self.inference(batched_inputs)

    def forward_tensor(self, img_tensors: torch.Tensor, hw=None) -> tuple:
        """modified_class.py:28-40: tensor in, 6-tuple of the FIRST image out; image_sizes fixed like FakeImageList."""
        assert not self.training
        features = self.backbone(img_tensors)
        images = FakeImageList(img_tensors, hw)
        proposals, _ = self.proposal_generator(images, features, None)
        results, _ = self.roi_heads(images, features, proposals, None)
        return flatten_to_tuple(results[0])


if "GeneralizedRCNN" not in META_ARCH_REGISTRY:
    META_ARCH_REGISTRY.register(GeneralizedRCNN)


def build_model(cfg):
    """detectron2 build_model: META_ARCH_REGISTRY.get(cfg.MODEL.META_ARCHITECTURE)(cfg).to(cfg.MODEL.DEVICE)."""
    model = META_ARCH_REGISTRY.get(cfg.MODEL.META_ARCHITECTURE)(cfg)
    model.to(torch.device(cfg.MODEL.DEVICE))
    return model
