"""Golden vectors for the norm variants of the mask head and of the FPN (VERDICT r02 "missing" 2), from the REFERENCE's own modules.

    python tests/golden/make_golden_norms.py      # needs /root/reference; writes tests/golden/norm_variants.pt

* SpatialAttentionMaskHead (sam.py:31-97) with MODEL.ROI_MASK_HEAD.NORM "GN" and "FrozenBN": the reference's own module, randomly initialised
  under a seed; the fixture holds its state dict (the parameter NAMES a checkpoint of that recipe carries), an input and the output logits.
* build_fcos_vovnet_fpn_backbone (vovnet.py:527-555) with MODEL.FPN.NORM "GN" and "FrozenBN": the reference's builder; the FPN class and
  get_norm are detectron2's and enter through tests/golden/d2_stub.py ("parity unpinned" against a real detectron2, as for the plain FPN).
  Conv weights are the V-39 synthetic ones (biases dropped: d2's FPN has none under a norm); the fixture holds the norm parameters only.
Data only.
"""
import os
import sys
from collections import OrderedDict

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402  (installs the d2 stand-ins and imports the reference package)

S = G.S
from detectron2.modeling.backbone.build import BACKBONE_REGISTRY  # noqa: E402
from centermask.modeling.centermask.sam import SpatialAttentionMaskHead  # noqa: E402
from centermask2_amd.structures import ShapeSpec  # noqa: E402


def ref_cfg(extra):
    cfg = G.ref_get_cfg()
    cfg.merge_from_file("/root/reference/centermask2/configs/centermask/zy_model_config.yaml")
    cfg.merge_from_list(["MODEL.DEVICE", "cpu"] + extra)
    cfg.freeze()
    return cfg


def randomise_norms(module, gen):
    for name, buf in list(module.named_parameters()) + list(module.named_buffers()):
        if ".norm." not in "." + name:
            continue
        with torch.no_grad():
            if name.endswith("running_var"):
                buf.copy_(torch.rand(buf.shape, generator=gen) + 0.5)
            elif name.endswith("weight"):
                buf.copy_(torch.rand(buf.shape, generator=gen) + 0.5)
            elif name.endswith("num_batches_tracked"):
                continue
            else:
                buf.copy_(torch.randn(buf.shape, generator=gen) * 0.1)


def main():
    out = {}
    for norm in ("GN", "FrozenBN"):
        torch.manual_seed(11)
        gen = torch.Generator().manual_seed(12)
        # a narrow head (128 channels, 2 convs) keeps the fixture small; same module, same code path
        head = SpatialAttentionMaskHead(ref_cfg(["MODEL.ROI_MASK_HEAD.NORM", norm, "MODEL.ROI_MASK_HEAD.CONV_DIM", 128, "MODEL.ROI_MASK_HEAD.NUM_CONV", 2]),
                                        ShapeSpec(channels=128, width=14, height=14)).eval()
        randomise_norms(head, gen)
        with torch.no_grad():
            head.predictor.weight.mul_(300.0)                    # std 0.001 logits would hide errors
            x = torch.randn((3, 128, 14, 14), generator=gen)
            y = head(x)
        sd = OrderedDict((k, v.clone()) for k, v in head.state_dict().items())
        print("mask head", norm, "keys", [k for k in sd if "fcn1" in k], "logits absmax %.3f" % float(y.abs().max()))
        out["mask_head_" + norm] = dict(state_dict=sd, x=x, logits=y.clone())
    sd_syn = S.make_synthetic_state_dict("V-39-eSE", seed=0)
    for norm in ("GN", "FrozenBN"):
        gen = torch.Generator().manual_seed(13)
        bb = BACKBONE_REGISTRY.get("build_fcos_vovnet_fpn_backbone")(ref_cfg(["MODEL.FPN.NORM", norm]), ShapeSpec(channels=3)).eval()
        own = bb.state_dict()
        sub = OrderedDict()
        for k in own:
            full = "backbone." + k
            if full in sd_syn:
                sub[k] = sd_syn[full]
        missing, unexpected = bb.load_state_dict(sub, strict=False)
        assert not unexpected and all(".norm." in m for m in missing), missing
        randomise_norms(bb, gen)
        norms = OrderedDict((k, v.clone()) for k, v in bb.state_dict().items() if ".norm." in k and k.startswith("fpn_"))
        assert not any(k.endswith(".bias") and ".norm." not in k and k.startswith("fpn_") for k in own), "d2 FPN has no conv bias under a norm"
        x = S.make_synthetic_images(1, 64, 96, seed0=821)
        with torch.no_grad():
            ref = bb(x)
        print("fpn", norm, "norm tensors", len(norms), {k: "%.3f" % float(v.abs().max()) for k, v in ref.items()})
        out["fpn_" + norm] = dict(norm_state=norms, keys=[k for k in own if k.startswith("fpn_") or k.startswith("top_block")], x=x,
                                  **{k: v.clone() for k, v in ref.items()})
    path = os.path.join(HERE, "norm_variants.pt")
    torch.save(out, path)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
