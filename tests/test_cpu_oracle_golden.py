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
    """V-19-slim-dw / V-19-dw (dw_conv3x3 + conv_reduction, vovnet.py:110-130,284-288), V-19-slim, V-19 and V-57
    (vovnet.py:50-88)."""
    g = golden("vovnet_bodies")
    assert set(g.keys()) == {"V-19-slim-dw-eSE", "V-19-dw-eSE", "V-19-slim-eSE", "V-19-eSE", "V-57-eSE"}
    for body, case in g.items():
        sd = S.make_synthetic_state_dict(body, 0)
        out = O.vovnet_forward(sd, case["x"], conv_body=body)
        for k in ("stage3", "stage4", "stage5"):
            close(out[k], case[k], 1e-6, body + " " + k)
        out = O.backbone_forward(sd, case["x32"], conv_body=body)
        for k in ("p3", "p4", "p5", "p6", "p7"):
            close(out[k], case[k], 1e-6, body + " " + k)


def test_v99_end_to_end_800x1280():
    """BASELINE config 5's body at its image size: the oracle against what the reference's own modules produced (one image)."""
    g = golden("e2e_v99_800x1280")
    sd = S.make_synthetic_state_dict("V-99-eSE", 0)
    x = S.make_synthetic_images(1, 800, 1280, seed0=int(g["image_seed0"]))
    res = O.centermask_inference(sd, x, [(800, 1280)], "V-99-eSE")[0]
    r = g["img0"]
    from .helpers import match_detections          # detections 9e-6 apart in score may swap between hosts (see its docstring)
    p = match_detections(res["scores"], res["classes"], res["locations"], r["scores"], r["classes"], r["locations"])
    close(res["boxes"], r["boxes"][p], 2e-5)
    close(res["scores"], r["scores"][p], 2e-5)
    close(res["pred_masks"], r["pred_masks"][p], 1e-4)
    close(res["mask_scores"], r["mask_scores"][p], 1e-4)


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


def test_pooler_by_area_and_roialign_v1():
    g, bb = golden("roi_crafted"), golden("backbone_small")
    feat, lev = O.roi_pooler([bb[k] for k in ("p3", "p4", "p5")], [g["img0"]["boxes"], g["img1"]["boxes"]], [(64, 96)] * 2, assign_crit="area",
                             aligned=False, canonical_box_size=40, canonical_level=4)
    assert torch.equal(lev, g["levels_area"])
    close(feat, g["roi_feat_area_v1"], 1e-6)
    # hand-derived: sqrt(area) == canonical size sits exactly on canonical_level; half the size is one level down; clamped at both ends
    b = torch.tensor([[0., 0., 224., 224.], [0., 0., 112., 112.], [0., 0., 111.9, 112.], [0., 0., 4., 4.], [0., 0., 4000., 4000.]])
    assert O.assign_boxes_to_levels(b, 2, 5, 224, 4).tolist() == [2, 1, 0, 0, 3]


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


# ---- hand-derived vectors for the third-party arithmetic (torchvision roi_align / nms, d2 batched_nms) ------------------------------
# These do not come from any implementation: each expectation follows from the published definition by hand, so they pin the oracle
# (and through tests/golden/d2_stub.py's independent restatement, the fixtures) without the oracle checking itself.
def test_roi_align_constant_map_gives_constant_output():
    f = torch.full((1, 3, 20, 30), 2.5)
    out = O.roi_align(f, torch.tensor([[0., 16., 24., 200., 130.], [0., 40.25, 17.5, 41.0, 150.]]), 1 / 8, 14, 0, True)
    assert float((out - 2.5).abs().max()) == 0.0                      # every sample lies inside the map: mean of equal values


def test_roi_align_linear_ramp_is_exact_at_bin_centres():
    """Bilinear interpolation reproduces a linear function, and the mean over a bin's regular sample grid is the value at the bin
    centre: out[ph, pw] = f(y0 + (ph + .5) bh, x0 + (pw + .5) bw) with y0 = y1*s - .5 (aligned), bh = (y2 - y1) s / P."""
    H, W, P, s = 24, 40, 14, 1 / 8
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    f = (0.25 * xx - 0.5 * yy + 3.0)[None, None]
    x1, y1, x2, y2 = 48.0, 40.0, 272.0, 152.0                          # in feature pixels: x 5.5..33.5, y 4.5..18.5 — all samples interior
    out = O.roi_align(f, torch.tensor([[0., x1, y1, x2, y2]]), s, P, 0, True)[0, 0]
    bh, bw = (y2 - y1) * s / P, (x2 - x1) * s / P
    ph = torch.arange(P, dtype=torch.float32)
    cy, cx = y1 * s - 0.5 + (ph + 0.5) * bh, x1 * s - 0.5 + (ph + 0.5) * bw
    want = 0.25 * cx[None, :] - 0.5 * cy[:, None] + 3.0
    assert float((out - want).abs().max()) < 2e-6


def test_roi_align_validity_boundary_at_minus_one_and_size():
    """torchvision: a sample with y < -1 or y > H contributes 0; y == -1 and y == H are still valid (clamped to the edge pixel)."""
    H, W = 4, 4
    f = torch.full((1, 1, H, W), 7.0)
    # P = 1, sampling_ratio 1: the single sample sits at the box centre.  aligned -> centre = (c1 + c2)/2 * s - 0.5 with s = 1.
    def centre_value(cy, cx):
        return float(O.roi_align(f, torch.tensor([[0., cx + 0.5 - 1.0, cy + 0.5 - 1.0, cx + 0.5 + 1.0, cy + 0.5 + 1.0]]), 1.0, 1, 1, True)[0, 0, 0, 0])
    assert centre_value(-1.0, 1.5) == 7.0                               # exactly -1: valid, clamped to row 0
    assert centre_value(-1.0 - 2 ** -10, 1.5) == 0.0                    # just outside
    assert centre_value(float(H), 1.5) == 7.0                           # exactly H: valid, clamped to row H-1
    assert centre_value(H + 2 ** -10, 1.5) == 0.0
    assert centre_value(1.5, -1.0) == 7.0 and centre_value(1.5, float(W)) == 7.0
    assert centre_value(1.5, W + 2 ** -10) == 0.0
    # the top edge: y in (H-1, H] interpolates between row H-1 and itself
    g = torch.arange(16, dtype=torch.float32).reshape(1, 1, 4, 4)
    v = float(O.roi_align(g, torch.tensor([[0., 0.5, 3.25 + 0.5 - 1.0, 2.5, 3.25 + 0.5 + 1.0]]), 1.0, 1, 1, True)[0, 0, 0, 0])
    assert v == 12.0 + 1.0                                              # sample (y 3.25 -> row 3, x 1.0): g[3][1] = 13


def test_nms_threshold_is_strict_and_areas_have_no_plus_one():
    """IoU exactly at, above and below 0.6 against box A = [0,0,10,10] (area 100, no +1): B = [0,0,10,6] has IoU 60/100 = 0.6 ->
    kept (suppression needs IoU > thr); C = [0,0,10,6.25] has 62.5/100 -> dropped; D = [0,0,10,5.75] -> kept."""
    A = [0., 0., 10., 10.]
    for other, kept in (([0., 0., 10., 6.], True), ([0., 0., 10., 6.25], False), ([0., 0., 10., 5.75], True)):
        boxes = torch.tensor([A, other])
        keep = O.batched_nms(boxes, torch.tensor([0.9, 0.8]), torch.tensor([3, 3]), 0.6)
        assert keep.tolist() == ([0, 1] if kept else [0]), (other, keep.tolist())
    # different labels never suppress each other; output is in descending score order
    boxes = torch.tensor([A, A, A])
    assert O.batched_nms(boxes, torch.tensor([0.5, 0.7, 0.6]), torch.tensor([1, 2, 3]), 0.6).tolist() == [1, 2, 0]
    # the same label: only the best survives
    assert O.batched_nms(boxes, torch.tensor([0.5, 0.7, 0.6]), torch.tensor([2, 2, 2]), 0.6).tolist() == [1]


def test_batched_nms_coordinate_trick_rounds_in_float32_at_label_79():
    """d2/torchvision shift every box by label * (max coordinate + 1) and run ONE nms in float32.  At label 79 with coordinates up
    to ~1300 the shift is ~1e5, where float32 has a spacing of 2^-7: the IoU is that of the ROUNDED boxes.  Expectation derived
    with explicit float32 arithmetic below, not from an implementation."""
    import numpy as np
    a = np.array([100.03, 200.02, 130.04, 240.01], dtype=np.float32)
    b = np.array([100.03, 200.02, 130.04, 224.012], dtype=np.float32)           # IoU with a = 0.6000x before the shift
    far = np.array([1290.0, 790.0, 1300.0, 800.0], dtype=np.float32)              # sets max coordinate = 1300
    shift = np.float32(79) * (np.float32(1300.0) + np.float32(1))
    ra, rb = a + shift, b + shift                                                 # float32 adds: rounded to multiples of 2^-7
    def iou(p, q):
        iw = max(np.float32(0), min(p[2], q[2]) - max(p[0], q[0])); ih = max(np.float32(0), min(p[3], q[3]) - max(p[1], q[1]))
        inter = np.float32(iw * ih)
        return inter / (np.float32((p[2] - p[0]) * (p[3] - p[1])) + np.float32((q[2] - q[0]) * (q[3] - q[1])) - inter)
    expect_b_kept = not (iou(ra, rb) > np.float32(0.6))
    assert (iou(a, b) > np.float32(0.6)) != (iou(ra, rb) > np.float32(0.6)), "the case must sit where the rounding decides"
    boxes = torch.from_numpy(np.stack([a, b, far]))
    keep = O.batched_nms(boxes, torch.tensor([0.9, 0.8, 0.1]), torch.tensor([79, 79, 0]), 0.6)
    assert keep.tolist() == ([0, 1, 2] if expect_b_kept else [0, 2])
