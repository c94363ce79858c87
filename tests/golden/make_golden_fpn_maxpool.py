"""Golden vectors for build_vovnet_fpn_backbone (vovnet.py:504-524): the REFERENCE's own builder — VoVNet V-39 + FPN with d2's
LastLevelMaxPool on top — on one seeded 64x96 image (odd p5: 2x3, so the stride-2 pick is pinned on an odd width).

    python tests/golden/make_golden_fpn_maxpool.py      # needs /root/reference; writes tests/golden/vovnet_fpn_maxpool.pt

The FPN and LastLevelMaxPool are detectron2's (absent offline): they enter through tests/golden/d2_stub.py, as for the FCOS backbone —
"parity unpinned" against a real detectron2; the VoVNet body and the builder are the reference's.  The backbone weights are the V-39
synthetic ones (the FPN lateral/output convs have the same names as in the FCOS backbone; the top block has no parameters).  Data only.
"""
import os
import sys
from collections import OrderedDict

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402  (installs the d2 stand-ins and imports the reference package)

S = G.S
from detectron2.modeling.backbone.build import BACKBONE_REGISTRY  # noqa: E402  (stub registry, filled by the reference)
from centermask2_amd.structures import ShapeSpec  # noqa: E402


def main():
    cfg = G.ref_get_cfg()
    cfg.merge_from_file("/root/reference/centermask2/configs/centermask/zy_model_config.yaml")
    cfg.merge_from_list(["MODEL.DEVICE", "cpu", "MODEL.BACKBONE.NAME", "build_vovnet_fpn_backbone"])
    cfg.freeze()
    backbone = BACKBONE_REGISTRY.get(cfg.MODEL.BACKBONE.NAME)(cfg, ShapeSpec(channels=3)).eval()
    sd = S.make_synthetic_state_dict("V-39-eSE", seed=0)
    sub = OrderedDict((k[len("backbone."):], v) for k, v in sd.items() if k.startswith("backbone.") and not k.startswith("backbone.top_block."))
    missing, unexpected = backbone.load_state_dict(sub, strict=True)
    assert not missing and not unexpected
    x = S.make_synthetic_images(1, 64, 96, seed0=811)
    with torch.no_grad():
        ref = backbone(x)
    assert list(ref.keys()) == ["p3", "p4", "p5", "p6"], list(ref.keys())
    assert torch.equal(ref["p6"], ref["p5"][:, :, ::2, ::2])
    out = dict(x=x, keys=list(sub.keys()), **{k: v.clone() for k, v in ref.items()})
    shp = backbone.output_shape()
    out["strides"] = torch.tensor([shp[k].stride for k in ref])
    for k, v in ref.items():
        print(k, tuple(v.shape), "stride", shp[k].stride, "absmax %.3f" % float(v.abs().max()))
    path = os.path.join(HERE, "vovnet_fpn_maxpool.pt")
    torch.save(out, path)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
