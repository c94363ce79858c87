"""not gpu: the oracle (CPU restatement) against the fixtures produced by the reference's own modules."""
import torch

from centermask2_amd import synthetic as S
from oracle import centermask_oracle as O

from .helpers import close, golden


def _sd():
    return S.make_synthetic_state_dict("V-39-eSE", 0)


def test_state_dict_keys_equal_reference_listing():
    import os
    from .helpers import GOLDEN
    keys = open(os.path.join(GOLDEN, "state_dict_keys_V39.txt")).read().split()
    assert set(keys) == set(S.model_param_shapes("V-39-eSE").keys()) and len(keys) == 293


def test_vovnet_odd_size_ceil_mode():
    g = golden("vovnet_odd")
    out = O.vovnet_forward(_sd(), g["x"])
    for k in ("stage3", "stage4", "stage5"):
        close(out[k], g[k], 1e-6, k)


def test_backbone_fpn_small():
    g = golden("backbone_small")
    out = O.backbone_forward(_sd(), g["x"])
    for k in ("p3", "p4", "p5", "p6", "p7"):
        close(out[k], g[k], 1e-6, k)


def test_other_vovnet_bodies_incl_depthwise():
    """V-19-slim-dw / V-19-dw (dw_conv3x3 + conv_reduction, vovnet.py:110-130,284-288) and V-19-slim."""
    g = golden("vovnet_bodies")
    assert set(g.keys()) == {"V-19-slim-dw-eSE", "V-19-dw-eSE", "V-19-slim-eSE"}
    for body, case in g.items():
        sd = S.make_synthetic_state_dict(body, 0)
        out = O.vovnet_forward(sd, case["x"], conv_body=body)
        for k in ("stage3", "stage4", "stage5"):
            close(out[k], case[k], 1e-6, body + " " + k)
        out = O.backbone_forward(sd, case["x32"], conv_body=body)
        for k in ("p3", "p4", "p5", "p6", "p7"):
            close(out[k], case[k], 1e-6, body + " " + k)


def test_fcos_head_decode_nms_small():
    g, bb = golden("fcos_small"), golden("backbone_small")
    sd = dict(_sd())
    key = "proposal_generator.fcos_head.cls_logits.bias"
    sd[key] = sd[key] + float(g["cls_bias_shift"])
    lg, reg, ctr = O.fcos_head_forward(sd, [bb[k] for k in ("p3", "p4", "p5", "p6", "p7")])
    for l in range(5):
        close(lg[l], g["logits"][l], 1e-6)
        close(reg[l], g["reg"][l], 1e-6)
        close(ctr[l], g["ctr"][l], 1e-6)
    props, cands = O.fcos_predict_proposals(g["logits"], g["reg"], g["ctr"], return_candidates=True)
    for i in range(2):
        r = g["img{}".format(i)]
        assert cands[i]["scores"].shape[0] == int(r["num_candidates"])
        assert torch.equal(props[i]["classes"], r["classes"]) and torch.equal(props[i]["locations"], r["locations"])
        close(props[i]["boxes"], r["boxes"], 1e-7)
        close(props[i]["scores"], r["scores"], 1e-7)


def test_roi_heads_crafted_boxes():
    g, bb = golden("roi_crafted"), golden("backbone_small")
    props = [dict(boxes=g["img{}".format(i)]["boxes"], classes=g["img{}".format(i)]["classes"], scores=g["img{}".format(i)]["scores"],
                  locations=g["img{}".format(i)]["boxes"][:, :2]) for i in range(2)]
    res, inter = O.roi_heads_forward(_sd(), bb, props, [(64, 96)] * 2, return_intermediates=True)
    assert torch.equal(inter["levels"], g["levels"])
    close(inter["roi_feat"], g["roi_feat"], 1e-6)
    close(inter["mask_logits"], g["mask_logits"], 1e-6)
    close(inter["maskiou"], g["maskiou"], 1e-6)
    for i in range(2):
        close(res[i]["pred_masks"], g["img{}".format(i)]["pred_masks"], 1e-6)
        close(res[i]["mask_scores"], g["img{}".format(i)]["mask_scores"], 1e-6)


def test_nms_and_level_edge_cases():
    # empty input, single box, duplicate boxes with tied scores (stable: lower index wins)
    e = O.batched_nms(torch.zeros((0, 4)), torch.zeros((0,)), torch.zeros((0,), dtype=torch.int64), 0.6)
    assert e.numel() == 0
    b = torch.tensor([[0., 0., 10., 10.], [0., 0., 10., 10.], [0., 0., 10., 10.], [20., 20., 30., 30.]])
    keep = O.batched_nms(b, torch.tensor([0.5, 0.5, 0.9, 0.5]), torch.tensor([1, 1, 2, 1]), 0.6)
    assert keep.tolist() == [2, 0, 3]
    # ratio rule: box area == img/4 -> level 3 (index 0); slightly larger -> index 1; zero area -> index 0; huge -> 2
    img = torch.full((4,), 64.0 * 96.0)
    s4 = (64 * 96 / 4) ** 0.5
    boxes = torch.tensor([[0, 0, s4, s4], [0, 0, s4 + 0.01, s4 + 0.01], [5, 5, 5, 5], [-500, -400, 700, 500]], dtype=torch.float32)
    assert O.assign_boxes_to_levels_by_ratio(boxes, img).tolist() == [0, 1, 0, 2]
    # roi_align of an empty roi set and of a zero-size roi
    f = torch.arange(2 * 4 * 6 * 8, dtype=torch.float32).reshape(2, 4, 6, 8)
    assert O.roi_align(f, torch.zeros((0, 5)), 0.5, 14, 0, True).shape == (0, 4, 14, 14)
    z = O.roi_align(f, torch.tensor([[1., 3., 3., 3., 3.]]), 1.0, 2, 0, True)
    assert float(z.abs().max()) == 0.0      # grid = ceil(0) = 0 samples -> 0 (torchvision behaviour)
