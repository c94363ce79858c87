"""-m gpu: the plugins end to end against the fixtures the reference's own modules produced (tests/golden).
Tolerances: features, logits, regression, centerness, mask probabilities and scores within 1e-3 ABSOLUTE (north_star "within 1e-3
fp32"; helpers.close_abs, no max|ref| factor; the observed maxima are printed at the end of the run), boxes in pixels within
2e-5 x max|coordinate|, labels/locations exact."""
import pytest
import torch

from .helpers import build_gpu_model, close, close_abs, golden, match_detections

pytestmark = pytest.mark.gpu


# 0 = the reference's exact order of detections (labels and locations compared with torch.equal, rank by rank).  Image 0 of the fixtures
# holds two detections 2.3e-6 apart (ranks 8 and 9); since the eSE gate is accumulated in float64 the HIP path
# reproduces the reference's order of them on all 8 + 2 fixture images.  A positive value (experiments only) would allow rank swaps
# between detections whose scores differ by less than it (helpers.match_detections).
ORDER_TOL = float(__import__("os").environ.get("CMK_TEST_ORDER_TOL", "0"))


@pytest.fixture(scope="module")
def model(dev):
    return build_gpu_model()[0]


def test_vovnet_odd_size_matches_reference(dev, model):
    g = golden("vovnet_odd")
    out = model.backbone.bottom_up(g["x"].to(dev))
    torch.cuda.synchronize()
    for k in ("stage3", "stage4", "stage5"):
        assert tuple(out[k].shape) == tuple(g[k].shape)
        close_abs(out[k], g[k], 1e-3, "features " + k)


def test_backbone_fpn_matches_reference(dev, model):
    g = golden("backbone_small")
    out = model.backbone(g["x"].to(dev))
    torch.cuda.synchronize()
    assert list(out.keys()) == ["p3", "p4", "p5", "p6", "p7"]
    for k in out:
        close_abs(out[k], g[k], 1e-3, "features " + k)
    shp = model.backbone.output_shape()
    assert shp["p3"].stride == 8 and shp["p7"].stride == 128 and shp["p5"].channels == 256 and model.backbone.size_divisibility == 32


def test_vovnet_fpn_maxpool_backbone_matches_reference(dev):
    """build_vovnet_fpn_backbone (vovnet.py:504-524: FPN + d2 LastLevelMaxPool) against what the reference's own builder produced
    (tests/golden/make_golden_fpn_maxpool.py); p6 is every second pixel of p5, bit for bit."""
    from centermask2_amd import synthetic as S
    from centermask2_amd.config import get_cfg, config_path
    from centermask2_amd.registry import BACKBONE_REGISTRY
    from centermask2_amd.structures import ShapeSpec
    g = golden("vovnet_fpn_maxpool")
    cfg = get_cfg()
    cfg.merge_from_file(config_path("centermask_V_39_eSE_FPN_ms_3x.yaml"))
    cfg.merge_from_list(["MODEL.DEVICE", "cuda", "MODEL.BACKBONE.NAME", "build_vovnet_fpn_backbone"])
    bb = BACKBONE_REGISTRY.get(cfg.MODEL.BACKBONE.NAME)(cfg, ShapeSpec(channels=3)).eval()
    sd = S.make_synthetic_state_dict("V-39-eSE", 0)
    bb.load_state_dict({k[len("backbone."):]: v for k, v in sd.items() if k.startswith("backbone.") and not k.startswith("backbone.top_block.")})
    out = bb.to(dev)(g["x"].to(dev))
    torch.cuda.synchronize()
    assert list(out.keys()) == ["p3", "p4", "p5", "p6"]
    for k in out:
        close_abs(out[k], g[k], 1e-3, "fpn+maxpool features " + k)
    assert torch.equal(out["p6"], out["p5"][:, :, ::2, ::2])


@pytest.mark.parametrize("body", ["V-19-slim-dw-eSE", "V-19-dw-eSE", "V-19-slim-eSE", "V-19-eSE", "V-57-eSE"])
def test_other_vovnet_bodies_match_reference(dev, body):
    """the depth-wise bodies (dw 3x3 kernel + 1x1, conv_reduction), the slim channel counts, V-19 and V-57
    (vovnet.py:50-88) — fixtures produced by the reference's own VoVNet/FPN (tests/golden/make_golden_bodies.py)."""
    g = golden("vovnet_bodies")[body]
    m = build_gpu_model(body)[0]
    out = m.backbone.bottom_up(g["x"].to(dev))
    torch.cuda.synchronize()
    for k in ("stage3", "stage4", "stage5"):
        assert tuple(out[k].shape) == tuple(g[k].shape)
        close_abs(out[k], g[k], 1e-3, "features " + body + " " + k)
    out = m.backbone(g["x32"].to(dev))
    torch.cuda.synchronize()
    for k in ("p3", "p4", "p5", "p6", "p7"):
        close_abs(out[k], g[k], 1e-3, "features " + body + " " + k)


def test_fcos_head_matches_reference(dev, model):
    g, bb = golden("fcos_small"), golden("backbone_small")
    feats = [bb[k].to(dev) for k in ("p3", "p4", "p5", "p6", "p7")]
    head = model.proposal_generator.fcos_head
    shift = float(g["cls_bias_shift"])
    head.cls_logits.bias.data += shift
    head.invalidate_packed()
    try:
        lg, reg, ctr, _ = head(feats)
        torch.cuda.synchronize()
    finally:
        head.cls_logits.bias.data -= shift
        head.invalidate_packed()
    for l in range(5):
        close_abs(lg[l], g["logits"][l], 1e-3, "fcos_small logits")
        close_abs(reg[l], g["reg"][l], 1e-3, "fcos_small reg")
        close_abs(ctr[l], g["ctr"][l], 1e-3, "fcos_small ctr")


def _probe_check(t_nchw, p, tol, what):
    flat = t_nchw.contiguous().reshape(-1).cpu()
    assert tuple(t_nchw.shape) == tuple(p["shape"].tolist()), what
    close_abs(flat[p["idx"]], p["val"], tol, what + " probe")
    close_abs(flat.double().mean().float().reshape(1), p["mean"].reshape(1), tol, what + " mean")


def _check_against_reference_image(inst, r, what, order_tol):
    """One image's Instances against the reference's tuple for it: the same detections, labels and ROI locations exact (order up to
    `order_tol` of score, see match_detections; 0 = identical order), scores / masks / mask scores within 1e-3 absolute."""
    assert len(inst) == r["scores"].shape[0], (what, len(inst), r["scores"].shape[0])
    if order_tol > 0:
        pg = match_detections(r["scores"], r["classes"], r["locations"], inst.scores, inst.pred_classes, inst.locations, tol=order_tol).to(inst.scores.device)
    else:
        pg = torch.arange(len(inst), device=inst.scores.device)
    assert torch.equal(inst.pred_classes[pg].cpu(), r["classes"]), what + ": labels differ"
    assert torch.equal(inst.locations[pg].cpu(), r["locations"]), what + ": ROI locations differ"
    close(inst.pred_boxes.tensor[pg], r["boxes"], 2e-5, "e2e boxes (pixels)")   # reg (1e-3 abs, probed) x stride
    close_abs(inst.scores[pg], r["scores"], 1e-3, "e2e scores")
    close_abs(inst.pred_masks[pg], r["pred_masks"], 1e-3, "e2e pred_masks")
    close_abs(inst.mask_scores[pg], r["mask_scores"], 1e-3, "e2e mask_scores")
    return pg


def test_end_to_end_800x1280_batch8_tuned_variants_match_reference(dev, model):
    """The benchmark configuration itself (8 x 3x800x1280 with the shipped variant table, i.e. the kernels bench.py times) against
    what the REFERENCE's own modules produced for all eight images (tests/golden/make_golden_bench8.py, tester.py:94-104 call
    order): labels and ROI locations exact on every image, feature / logit probes, scores, masks and mask scores within 1e-3
    absolute, candidate counts within the handful of scores that straddle 0.05 at fp32 noise."""
    from centermask2_amd import ops, synthetic as S
    from .helpers import load_shipped_variant_table
    g = golden("e2e_bench8_800x1280")
    B = int(g["num_images"])
    n_loaded = load_shipped_variant_table()
    try:
        x = S.make_synthetic_images(B, 800, 1280, seed0=int(g["image_seed0"])).to(dev)
        sizes = [(800, 1280)] * B
        out = model.inference_padded(x, sizes)
        feats = model.backbone(x)
        lg, reg, ctr, _ = model.proposal_generator.fcos_head([feats[k] for k in ("p3", "p4", "p5", "p6", "p7")])
        torch.cuda.synchronize()
        res = model.results_from_padded(out, sizes)
        cand = out["cand_counts"].cpu().tolist()
        for i in range(B):
            r = g["img{}".format(i)]
            for k in ("p3", "p4", "p5", "p6", "p7"):
                _probe_check(feats[k][i:i + 1], r[k], 1e-3, "bench8 " + k)
            for l in range(5):
                _probe_check(lg[l][i:i + 1], r["logits{}".format(l)], 1e-3, "bench8 logits")
                _probe_check(reg[l][i:i + 1], r["reg{}".format(l)], 1e-3, "bench8 reg")
                _probe_check(ctr[l][i:i + 1], r["ctr{}".format(l)], 1e-3, "bench8 ctr")
            # a candidate is sigmoid(logit) > 0.05 (fcos_outputs.py:412): a logit within fp32 noise of the threshold may fall on either side
            assert abs(cand[i] - int(r["num_candidates"])) <= 4, (i, cand[i], int(r["num_candidates"]))
            _check_against_reference_image(res[i], r, "bench8 image {}".format(i), ORDER_TOL)
    finally:
        ops._TUNED.clear()
    assert n_loaded >= 0


def test_end_to_end_800x1280_matches_reference(dev, model):
    """BASELINE config 1 shape: two 800x1280 images through backbone -> FCOS -> CenterROIHeads; compares with what the
    reference produced for the same seeded weights/images."""
    from centermask2_amd import synthetic as S
    from centermask2_amd.structures import FakeImageList
    g = golden("e2e_800x1280")
    x = S.make_synthetic_images(2, 800, 1280, seed0=int(g["image_seed0"])).to(dev)
    sizes = [(800, 1280), (800, 1280)]
    feats = model.backbone(x)
    for k in ("p3", "p4", "p5", "p6", "p7"):
        _probe_check(feats[k], g[k], 1e-3, k)
    lg, reg, ctr, _ = model.proposal_generator.fcos_head([feats[k] for k in ("p3", "p4", "p5", "p6", "p7")])
    for l in range(5):
        _probe_check(lg[l], g["logits{}".format(l)], 1e-3, "logits{}".format(l))
        _probe_check(reg[l], g["reg{}".format(l)], 1e-3, "reg{}".format(l))
        _probe_check(ctr[l], g["ctr{}".format(l)], 1e-3, "ctr{}".format(l))
    res = model.inference(FakeImageList(x, sizes), do_preprocess=False, do_postprocess=False)
    torch.cuda.synchronize()
    for i in range(2):
        r, inst = g["img{}".format(i)], res[i]
        _check_against_reference_image(inst, r, "e2e image {}".format(i), ORDER_TOL)
        assert inst.pred_classes.dtype == torch.int64 and tuple(inst.pred_masks.shape[1:]) == (1, 28, 28)
    t = model.forward_tensor(x[:1], hw=[(800, 1280)])
    assert [tuple(v.shape[1:]) for v in t] == [(2,), (), (4,), (), (1, 28, 28), ()]


def test_v99_small_image_backbone(dev):
    """V2-99-eSE (blocks [1,3,9,3], vovnet.py:90-98) through the same kernels at a small size against the oracle."""
    from centermask2_amd import synthetic as S
    from oracle import centermask_oracle as O
    model, sd = build_gpu_model("V-99-eSE")
    x = S.make_synthetic_images(1, 128, 192, seed0=555)
    feats = model.backbone(x.to(dev))
    ref = O.backbone_forward(sd, x, "V-99-eSE")
    for k in ("p3", "p4", "p5", "p6", "p7"):
        close_abs(feats[k], ref[k], 1e-3, "V-99 features " + k)


def test_config5_v99_batch8_800x1280(dev):
    """BASELINE config 5 at its single-GPU workload: V2-99-eSE, 8 x 3x800x1280 through inference_padded on the kernels the bench
    uses (the shipped measured variant table).  Image 0 against the fixture the reference's own modules produced
    (tests/golden/make_golden_v99.py), images 0 and 5 against the oracle: labels and ROI locations exact, boxes/scores as in the V-39
    test.  Masks and mask scores are checked with the ORACLE's boxes fed to roi_heads.forward_with_given_boxes, so box noise cannot
    move a bilinear sample across ROIAlign's validity edge: every ROI is compared, none excluded."""
    import os
    from centermask2_amd import ops, synthetic as S
    from centermask2_amd.structures import Boxes, Instances
    from oracle import centermask_oracle as O
    from .helpers import GOLDEN_ROOT
    g = golden("e2e_v99_800x1280")
    model, sd = build_gpu_model("V-99-eSE")
    table = os.path.join(os.path.dirname(GOLDEN_ROOT), "centermask2_amd", "tuned", "mi355x_V-99-eSE_b8_800x1280.json")
    saved = dict(ops._TUNED)
    try:
        assert os.path.exists(table) and ops.load_tuned(table) > 0
        B = 8
        x = S.make_synthetic_images(B, 800, 1280, seed0=int(g["image_seed0"]))
        sizes = [(800, 1280)] * B
        xd = x.to(dev)
        with torch.no_grad():
            out = model.inference_padded(xd, sizes)
            feats = model.backbone(xd)
        torch.cuda.synchronize()
        res = model.results_from_padded(out, sizes)
        assert max(out["cand_counts"].cpu().tolist()) <= model.proposal_generator.candidate_capacity
        refs = {0: g["img0"]}
        torch.set_num_threads(min(16, os.cpu_count() or 1))
        for i in (0, 5):
            want = O.centermask_inference(sd, x[i:i + 1], [(800, 1280)], "V-99-eSE")[0]
            if i == 0:      # the oracle on THIS host against the fixture made in the build container (order: see match_detections)
                r = refs[0]
                pr = match_detections(want["scores"], want["classes"], want["locations"], r["scores"], r["classes"], r["locations"])
                close(want["boxes"], r["boxes"][pr], 2e-5, "oracle vs reference boxes")
            inst = res[i]
            assert len(inst) == want["scores"].shape[0] == 50
            pg = match_detections(want["scores"], want["classes"], want["locations"], inst.scores, inst.pred_classes, inst.locations).to(dev)
            assert torch.equal(inst.pred_classes[pg].cpu(), want["classes"]) and torch.equal(inst.locations[pg].cpu(), want["locations"])
            close(inst.pred_boxes.tensor[pg], want["boxes"], 2e-5, "boxes")
            close_abs(inst.scores[pg], want["scores"], 1e-4, "V-99 scores")
            # mask branch on the oracle's boxes (reference API: center_heads.py:413-444): every ROI compared, none excluded
            given = Instances((800, 1280))
            given.pred_boxes = Boxes(want["boxes"].to(dev))
            given.pred_classes = want["classes"].to(dev)
            given.scores = want["scores"].to(dev)
            fi = {k: v[i:i + 1] for k, v in feats.items()}
            got = model.roi_heads.forward_with_given_boxes(fi, [given])[0]
            torch.cuda.synchronize()
            close_abs(got.pred_masks, want["pred_masks"], 1e-3, "V-99 pred_masks on the oracle's boxes")
            close_abs(got.mask_scores, want["mask_scores"], 1e-3, "V-99 mask_scores on the oracle's boxes")
            if i == 0:
                close_abs(got.pred_masks, refs[0]["pred_masks"][pr], 1e-3, "V-99 pred_masks vs the reference fixture")
                close_abs(got.mask_scores, refs[0]["mask_scores"][pr], 1e-3, "V-99 mask_scores vs the reference fixture")
    finally:
        ops._TUNED.clear()
        ops._TUNED.update(saved)


def test_inference_with_pre_and_postprocess(dev, model):
    """GeneralizedRCNN.inference on raw uint8-valued images (tester.py:25-75 with do_preprocess/do_postprocess): the post-processed
    boxes are EXACTLY the raw boxes times (output size / network input size), clipped — d2's rule, not the 800/1333 rule of the
    deployment path — and images without height/width keys come back at the network input size, unscaled."""
    g = torch.Generator().manual_seed(77)
    imgs = [{"image": torch.randint(0, 256, (3, 200, 300), generator=g).float().to(dev), "height": 100, "width": 150},
            {"image": torch.randint(0, 256, (3, 180, 260), generator=g).to(torch.uint8).to(dev), "height": 90, "width": 195},
            {"image": torch.randint(0, 256, (3, 192, 256), generator=g).float().to(dev)}]
    out = model.inference(imgs)
    raw = model.inference(imgs, do_postprocess=False)
    torch.cuda.synchronize()
    assert len(out) == 3
    for o, r, im in zip(out, raw, imgs):
        inst = o["instances"]
        h_in, w_in = im["image"].shape[-2:]
        h, w = im.get("height", h_in), im.get("width", w_in)
        assert inst.image_size == (h, w) and r.image_size == (h_in, w_in)
        want = r.pred_boxes.tensor.clone()
        want[:, 0::2] = (want[:, 0::2] * (w / w_in)).clamp(0, w)
        want[:, 1::2] = (want[:, 1::2] * (h / h_in)).clamp(0, h)
        keep = ((want[:, 2] - want[:, 0]) > 0) & ((want[:, 3] - want[:, 1]) > 0)
        assert len(inst) == int(keep.sum())
        assert torch.equal(inst.pred_boxes.tensor, want[keep]), "boxes must be the raw boxes x (output / input size)"
        assert torch.equal(inst.scores, r.scores[keep]) and torch.equal(inst.pred_classes, r.pred_classes[keep])
        if len(inst):
            assert inst.pred_masks.dtype == torch.bool and tuple(inst.pred_masks.shape[1:]) == (h, w)
    assert len(raw[0]) > 0


def test_zero_detections_and_capacity_overflow(dev):
    """Edge cases of the detection tail: no score above 0.05 anywhere (empty Instances with the right field shapes, the ROI
    heads run on zero valid slots), and more candidates than the workspace capacity (re-run with a larger one, never truncated)."""
    from centermask2_amd import synthetic as S
    from centermask2_amd.structures import FakeImageList
    model, _ = build_gpu_model()
    x = S.make_synthetic_images(2, 128, 160, seed0=99).to(dev)
    images = FakeImageList(x, [(128, 160), (100, 150)])
    head = model.proposal_generator.fcos_head
    head.cls_logits.bias.data -= 30.0
    head.invalidate_packed()
    try:
        res = model.inference(images, do_preprocess=False, do_postprocess=False)
        torch.cuda.synchronize()
        for inst in res:
            assert len(inst) == 0
            assert tuple(inst.pred_boxes.tensor.shape) == (0, 4) and tuple(inst.pred_masks.shape) == (0, 1, 28, 28)
            assert tuple(inst.mask_scores.shape) == (0,) and inst.pred_classes.dtype == torch.int64
    finally:
        head.cls_logits.bias.data += 30.0
        head.invalidate_packed()
    # more candidates than the capacity: the reference is unbounded (fcos_outputs.py:444-449), so the tail is re-run with a
    # capacity sized from the true count and the results equal those of a run whose capacity was large from the start
    head.cls_logits.bias.data += 4.0
    head.invalidate_packed()
    fcos = model.proposal_generator
    try:
        want = model.inference(images, do_preprocess=False, do_postprocess=False)
        torch.cuda.synchronize()
        assert fcos.candidate_capacity == 131072
        fcos.candidate_capacity = 1024
        out = model.inference_padded(x, images.image_sizes)
        assert bool(out["overflow"].any()) and int(out["cand_counts"].max()) > 1024          # visible on the padded path
        got = model.results_from_padded(out, images.image_sizes)                               # ... and resolved at the sync point
        assert fcos.candidate_capacity >= int(out["cand_counts"].max())
        fcos.candidate_capacity = 1024
        got2 = model.inference(images, do_preprocess=False, do_postprocess=False)
        torch.cuda.synchronize()
        for a, b, c in zip(want, got, got2):
            assert len(a) == len(b) == len(c) > 0
            for other in (b, c):
                assert torch.equal(a.pred_classes, other.pred_classes) and torch.equal(a.pred_boxes.tensor, other.pred_boxes.tensor)
                assert torch.equal(a.scores, other.scores) and torch.equal(a.pred_masks, other.pred_masks)
    finally:
        head.cls_logits.bias.data -= 4.0
        head.invalidate_packed()
        fcos.candidate_capacity = 131072


def test_thresh_with_ctr_model_matches_oracle(dev):
    """cfg MODEL.FCOS.THRESH_WITH_CTR True and POST_NMS_TOPK_TEST 100 through build_model (both used to be refused)."""
    from centermask2_amd import synthetic as S
    from centermask2_amd.config import get_cfg, config_path
    from centermask2_amd.modeling import build_model
    from oracle import centermask_oracle as O
    cfg = get_cfg()
    cfg.merge_from_file(config_path("centermask_V_39_eSE_FPN_ms_3x.yaml"))
    cfg.merge_from_list(["MODEL.DEVICE", "cuda", "MODEL.FCOS.THRESH_WITH_CTR", True, "MODEL.FCOS.POST_NMS_TOPK_TEST", 100])
    cfg.freeze()
    sd = S.make_synthetic_state_dict("V-39-eSE", 0)
    m = build_model(cfg).eval()
    m.load_state_dict(sd)
    x = S.make_synthetic_images(1, 256, 320, seed0=4321)
    feats = m.backbone(x.to(dev))
    det, _ = m.proposal_generator.forward_padded(feats)
    torch.cuda.synchronize()
    of = O.backbone_forward(sd, x)
    lg, reg, ctr = O.fcos_head_forward(sd, [of[k] for k in ("p3", "p4", "p5", "p6", "p7")])
    want = O.fcos_predict_proposals(lg, reg, ctr, post_nms_topk=100, thresh_with_ctr=True)[0]
    k = int(det["counts"][0])
    assert k == want["scores"].shape[0] and k > 64
    assert torch.equal(det["cls"][0, :k].cpu(), want["classes"]) and torch.equal(det["loc"][0, :k].cpu(), want["locations"])
    close_abs(det["score"][0, :k], want["scores"], 1e-4, "thresh_with_ctr scores")


@pytest.mark.parametrize("norm", ["GN", "FrozenBN"])
def test_mask_head_norm_variants_match_reference(dev, norm):
    """MODEL.ROI_MASK_HEAD.NORM (sam.py:53,66): the reference's own SpatialAttentionMaskHead with "GN" / "FrozenBN" convs (bias-free conv ->
    norm -> ReLU), its state dict loaded by name into ours (tests/golden/make_golden_norms.py); all-class mask logits within 1e-3 absolute."""
    from centermask2_amd.config import get_cfg, config_path
    from centermask2_amd.modeling.centermask.center_heads import SpatialAttentionMaskHead
    from centermask2_amd.structures import ShapeSpec
    g = golden("norm_variants")["mask_head_" + norm]
    cfg = get_cfg()
    cfg.merge_from_file(config_path("centermask_V_39_eSE_FPN_ms_3x.yaml"))
    cfg.merge_from_list(["MODEL.DEVICE", "cuda", "MODEL.ROI_MASK_HEAD.NORM", norm, "MODEL.ROI_MASK_HEAD.CONV_DIM", 128, "MODEL.ROI_MASK_HEAD.NUM_CONV", 2])
    head = SpatialAttentionMaskHead(cfg, ShapeSpec(channels=128, width=14, height=14)).eval()
    missing, unexpected = head.load_state_dict(g["state_dict"], strict=True)
    assert not missing and not unexpected
    y = head.to(dev)(g["x"].to(dev))
    torch.cuda.synchronize()
    close_abs(y, g["logits"], 1e-3, "mask head NORM " + norm + " logits")


@pytest.mark.parametrize("norm", ["GN", "FrozenBN"])
def test_fpn_norm_variants_match_reference(dev, norm):
    """MODEL.FPN.NORM (vovnet.py:550): the reference's backbone builder with a norm behind every FPN conv (d2's FPN and get_norm through the
    build container's stand-ins: unpinned against a real detectron2, like the plain FPN) — p3..p7 within 1e-3 absolute."""
    from centermask2_amd import synthetic as S
    from centermask2_amd.config import get_cfg, config_path
    from centermask2_amd.registry import BACKBONE_REGISTRY
    from centermask2_amd.structures import ShapeSpec
    g = golden("norm_variants")["fpn_" + norm]
    cfg = get_cfg()
    cfg.merge_from_file(config_path("centermask_V_39_eSE_FPN_ms_3x.yaml"))
    cfg.merge_from_list(["MODEL.DEVICE", "cuda", "MODEL.FPN.NORM", norm])
    bb = BACKBONE_REGISTRY.get(cfg.MODEL.BACKBONE.NAME)(cfg, ShapeSpec(channels=3)).eval()
    own = bb.state_dict()
    assert set(k for k in own if k.startswith("fpn_") or k.startswith("top_block")) == set(g["keys"])
    sd = S.make_synthetic_state_dict("V-39-eSE", 0)
    sub = {k: sd["backbone." + k] for k in own if "backbone." + k in sd}
    sub.update(g["norm_state"])
    missing, unexpected = bb.load_state_dict(sub, strict=False)
    assert not unexpected and all(m.endswith("num_batches_tracked") for m in missing), (missing, unexpected)
    out = bb.to(dev)(g["x"].to(dev))
    torch.cuda.synchronize()
    for k in ("p3", "p4", "p5", "p6", "p7"):
        close_abs(out[k], g[k], 1e-3, "FPN NORM " + norm + " " + k)


def split_variant_table(body="V-39-eSE"):
    """The shipped variant table with every conv the pointwise GEMM kernel runs without split-K (plain 1x1, 1x1 with the FPN top-down add, 3x3 in
    the gather form) moved to the opt-in bf16-split form (tune 10/32/4)."""
    import json, os
    from centermask2_amd import ops
    from .helpers import GOLDEN_ROOT
    path = os.path.join(os.path.dirname(GOLDEN_ROOT), "centermask2_amd", "tuned", "mi355x_{}_b8_800x1280.json".format(body))
    table = json.load(open(path))
    out = {}
    for k, v in table.items():
        if len(v) == 3 and ((k.startswith("k1s1") and v[0] == 8 and ("_res0_" in k or "_res2_" in k)) or (k.startswith("k3") and v[0] == 9 and "_res0_" in k)):
            v = [10, 32, 4]
        out[ops._str_to_key(k)] = tuple(v)
    return out


def test_end_to_end_batch8_with_the_opt_in_split_gemm(dev, monkeypatch):
    """OPT-IN path (ops.ALLOW_SPLIT_BF16; nothing selects it by default): the 1x1 aggregation convs, the mask head's deconv-as-1x1 as GEMMs
    whose fp32 products are rebuilt from bf16 pieces (cmk.h tune_wm 10).  The same gate as the default path: all eight bench images
    against the reference's fixture — labels, ROI locations and their ORDER exact, features / logits / scores / masks within 1e-3 absolute."""
    from centermask2_amd import ops, synthetic as S
    monkeypatch.setattr(ops, "ALLOW_SPLIT_BF16", True)
    model = build_gpu_model()[0]                      # packs the split weights
    g = golden("e2e_bench8_800x1280")
    B = int(g["num_images"])
    saved = dict(ops._TUNED)
    try:
        ops._TUNED.clear()
        ops._TUNED.update(split_variant_table())
        assert sum(1 for v in ops._TUNED.values() if v[0] == 10) >= 9
        x = S.make_synthetic_images(B, 800, 1280, seed0=int(g["image_seed0"])).to(dev)
        sizes = [(800, 1280)] * B
        out = model.inference_padded(x, sizes)
        feats = model.backbone(x)
        lg, reg, ctr, _ = model.proposal_generator.fcos_head([feats[k] for k in ("p3", "p4", "p5", "p6", "p7")])
        torch.cuda.synchronize()
        res = model.results_from_padded(out, sizes)
        for i in range(B):
            r = g["img{}".format(i)]
            for k in ("p3", "p4", "p5", "p6", "p7"):
                _probe_check(feats[k][i:i + 1], r[k], 1e-3, "split-gemm " + k)
            for l in range(5):
                _probe_check(lg[l][i:i + 1], r["logits{}".format(l)], 1e-3, "split-gemm logits")
                _probe_check(reg[l][i:i + 1], r["reg{}".format(l)], 1e-3, "split-gemm reg")
            _check_against_reference_image(res[i], r, "split-gemm image {}".format(i), ORDER_TOL)
    finally:
        ops._TUNED.clear()
        ops._TUNED.update(saved)


def test_end_to_end_batch8_with_the_opt_in_direct_split_convs(dev, monkeypatch):
    """Second OPT-IN level (ops.ALLOW_SPLIT_F16; nothing selects it by default): the pointwise GEMMs (tune_wm 12) and the 3x3 convs of
    stage 2 / 3, the FPN outputs, the FCOS towers (fused GroupNorm statistics and input affine) and the mask / mask-IoU heads run as direct
    implicit GEMMs on TWO fp16 pieces per fp32 operand (cmk.h tune_wm 11, conv_sp3.hip: 22-bit operands, three products, fp32 accumulation) —
    the measured table `tuned/mi355x_V-39-eSE_b8_800x1280_split3.json`.  The SAME gate as the default path on all eight bench images: labels,
    ROI locations and their ORDER exact (ORDER_TOL = 0), features / logits / regression / scores / masks within 1e-3 absolute."""
    import os
    from centermask2_amd import ops, synthetic as S
    from .helpers import GOLDEN_ROOT
    monkeypatch.setattr(ops, "ALLOW_SPLIT_BF16", True)
    monkeypatch.setattr(ops, "ALLOW_SPLIT_F16", True)
    model = build_gpu_model()[0]                      # packs the split weights
    g = golden("e2e_bench8_800x1280")
    B = int(g["num_images"])
    saved = dict(ops._TUNED)
    try:
        ops._TUNED.clear()
        n = ops.load_tuned(os.path.join(os.path.dirname(GOLDEN_ROOT), "centermask2_amd", "tuned", "mi355x_V-39-eSE_b8_800x1280_split3.json"))
        assert n > 0 and sum(1 for v in ops._TUNED.values() if v[0] == 11) >= 6 and sum(1 for v in ops._TUNED.values() if v[0] == 12) >= 9
        x = S.make_synthetic_images(B, 800, 1280, seed0=int(g["image_seed0"])).to(dev)
        sizes = [(800, 1280)] * B
        out = model.inference_padded(x, sizes)
        feats = model.backbone(x)
        lg, reg, ctr, _ = model.proposal_generator.fcos_head([feats[k] for k in ("p3", "p4", "p5", "p6", "p7")])
        torch.cuda.synchronize()
        res = model.results_from_padded(out, sizes)
        for i in range(B):
            r = g["img{}".format(i)]
            for k in ("p3", "p4", "p5", "p6", "p7"):
                _probe_check(feats[k][i:i + 1], r[k], 1e-3, "direct-split " + k)
            for l in range(5):
                _probe_check(lg[l][i:i + 1], r["logits{}".format(l)], 1e-3, "direct-split logits")
                _probe_check(reg[l][i:i + 1], r["reg{}".format(l)], 1e-3, "direct-split reg")
            _check_against_reference_image(res[i], r, "direct-split image {}".format(i), ORDER_TOL)
    finally:
        ops._TUNED.clear()
        ops._TUNED.update(saved)
