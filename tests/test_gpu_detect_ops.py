"""-m gpu: FCOS selection/decode, sort+NMS+top-k and the ROI-head kernels through the C ABI, against
(a) the fixtures the reference's own modules produced (tests/golden) and (b) the oracle on seeded inputs.
Bar: indices / labels / locations bit-exact; boxes, scores within 1e-6 rel; mask logits within 1e-3 (north_star)."""
import pytest
import torch

from centermask2_amd import ops
from centermask2_amd.ops import View
from oracle import centermask_oracle as O

from .helpers import close, close_abs, golden

pytestmark = pytest.mark.gpu
STRIDES = (8, 16, 32, 64, 128)


def _to_dev_levels(logits, reg, ctr, dev):
    lg = [t.permute(0, 2, 3, 1).contiguous().to(dev) for t in logits]
    rc = [torch.cat([r, c], 1).permute(0, 2, 3, 1).contiguous().to(dev) for r, c in zip(reg, ctr)]
    return lg, rc


def _run_detect(logits, reg, ctr, dev, cap=65536, topk=50, thr=0.05, nms=0.6, with_ctr=False):
    lg, rc = _to_dev_levels(logits, reg, ctr, dev)
    cand = ops.fcos_select(lg, rc, STRIDES[:len(lg)], thr, cap, with_ctr)
    det = ops.nms_topk(cand, nms, topk)
    torch.cuda.synchronize()
    return cand, det


def test_fcos_select_and_nms_match_reference_fixture(dev):
    g = golden("fcos_small")
    cand, det = _run_detect(g["logits"], g["reg"], g["ctr"], dev)
    for i in range(2):
        ref = g["img{}".format(i)]
        assert int(cand["counts"][i]) == int(ref["num_candidates"])
        k = int(det["counts"][i])
        assert k == ref["scores"].shape[0]
        assert torch.equal(det["cls"][i, :k].cpu(), ref["classes"])
        assert torch.equal(det["loc"][i, :k].cpu(), ref["locations"])
        close(det["box"][i, :k], ref["boxes"], 1e-6, "boxes")
        close(det["score"][i, :k], ref["scores"], 1e-6, "scores")


def test_candidates_match_oracle_order_and_values(dev):
    g = golden("fcos_small")
    cand, _ = _run_detect(g["logits"], g["reg"], g["ctr"], dev)
    _, ocands = O.fcos_predict_proposals(g["logits"], g["reg"], g["ctr"], return_candidates=True)
    for i in range(2):
        n = ocands[i]["scores"].shape[0]
        assert int(cand["counts"][i]) == n
        assert torch.equal(cand["cls"][i, :n].cpu().long(), ocands[i]["classes"])       # nonzero order: level, location, class
        assert torch.equal(cand["loc"][i, :n].cpu(), ocands[i]["locations"])
        close(cand["box"][i, :n], ocands[i]["boxes"], 1e-6, "cand boxes")
        close(cand["score"][i, :n], ocands[i]["scores"], 1e-6, "cand scores")


def _crafted_levels(n_img, shapes, seed, bias):
    g = torch.Generator().manual_seed(seed)
    logits = [torch.randn((n_img, 80, h, w), generator=g) * 1.2 + bias for h, w in shapes]
    reg = [torch.rand((n_img, 4, h, w), generator=g) * 6.0 for h, w in shapes]
    ctr = [torch.randn((n_img, 1, h, w), generator=g) for h, w in shapes]
    return logits, reg, ctr


@pytest.mark.parametrize("seed,bias", [(1, -4.5), (2, -3.0), (3, -9.0)])
def test_detect_random_levels_vs_oracle(dev, seed, bias):
    """Random heads incl. the empty case (bias -9: no candidate) and ragged level sizes."""
    shapes = [(25, 40), (13, 20), (7, 10), (4, 5), (2, 3)]
    logits, reg, ctr = _crafted_levels(2, shapes, seed, bias)
    cand, det = _run_detect(logits, reg, ctr, dev)
    oprops, ocands = O.fcos_predict_proposals(logits, reg, ctr, return_candidates=True)
    for i in range(2):
        assert int(cand["counts"][i]) == ocands[i]["scores"].shape[0]
        k = int(det["counts"][i])
        assert k == oprops[i]["scores"].shape[0]
        assert torch.equal(det["cls"][i, :k].cpu(), oprops[i]["classes"])
        assert torch.equal(det["loc"][i, :k].cpu(), oprops[i]["locations"])
        close(det["box"][i, :k], oprops[i]["boxes"], 1e-6)
        close(det["score"][i, :k], oprops[i]["scores"], 1e-6)


@pytest.mark.parametrize("topk", [50, 100, 300])
def test_thresh_with_ctr_and_large_topk_vs_oracle(dev, topk):
    """MODEL.FCOS.THRESH_WITH_CTR = True (fcos_outputs.py:410-420: the threshold applies to cls*ctr) and POST_NMS_TOPK_TEST beyond
    one wave (the fork's top-k is whatever the config says, :477-482)."""
    shapes = [(25, 40), (13, 20), (7, 10), (4, 5), (2, 3)]
    logits, reg, ctr = _crafted_levels(2, shapes, 11, -2.5)
    for with_ctr in (False, True):
        cand, det = _run_detect(logits, reg, ctr, dev, cap=131072, topk=topk, with_ctr=with_ctr)     # ~68k candidates per image: also the >= 40000 per-class NMS branch
        assert int(cand["counts"].max()) <= 131072
        oprops, ocands = O.fcos_predict_proposals(logits, reg, ctr, post_nms_topk=topk, return_candidates=True, thresh_with_ctr=with_ctr)
        for i in range(2):
            assert int(cand["counts"][i]) == ocands[i]["scores"].shape[0]
            k = int(det["counts"][i])
            assert k == oprops[i]["scores"].shape[0] and (topk == 50 or k > 64)
            assert torch.equal(det["cls"][i, :k].cpu(), oprops[i]["classes"])
            assert torch.equal(det["loc"][i, :k].cpu(), oprops[i]["locations"])
            close(det["box"][i, :k], oprops[i]["boxes"], 1e-6)
            close(det["score"][i, :k], oprops[i]["scores"], 1e-6)
    n_plain = O.fcos_predict_proposals(logits, reg, ctr, return_candidates=True)[1][0]["scores"].shape[0]
    n_ctr = O.fcos_predict_proposals(logits, reg, ctr, return_candidates=True, thresh_with_ctr=True)[1][0]["scores"].shape[0]
    assert n_ctr < n_plain            # the two rules really select different sets here


def test_nms_heavy_overlap_ties_and_threshold_edges(dev):
    """Same-class boxes stacked on one spot (long suppression chains), exact score ties (stable order = index order)
    and IoU pairs straddling 0.6."""
    n, cap = 1, 4096
    g = torch.Generator().manual_seed(5)
    m = 900
    base = torch.tensor([100.0, 120.0, 300.0, 360.0])
    box = base[None, :] + torch.randn((m, 4), generator=g) * 12.0
    box[:50] = base                       # identical boxes
    cls = torch.randint(0, 3, (m,), generator=g)
    score = torch.rand((m,), generator=g) * 0.5 + 0.2
    score[10:40] = 0.5                    # ties
    # pairs near the threshold: shift so IoU ~ 0.6 +- 1e-3
    box[60] = torch.tensor([0.0, 0.0, 100.0, 100.0]); box[61] = torch.tensor([0.0, 0.0, 100.0, 60.05]); cls[60] = cls[61] = 7
    box[62] = torch.tensor([500.0, 0.0, 600.0, 100.0]); box[63] = torch.tensor([500.0, 0.0, 600.0, 59.95]); cls[62] = cls[63] = 7
    score[60], score[61], score[62], score[63] = 0.99, 0.98, 0.97, 0.96
    cand = dict(box=torch.zeros((n, cap, 4)), score=torch.zeros((n, cap)), cls=torch.zeros((n, cap), dtype=torch.int32),
                loc=torch.zeros((n, cap, 2)), counts=torch.tensor([m], dtype=torch.int32))
    cand["box"][0, :m], cand["score"][0, :m], cand["cls"][0, :m] = box, score, cls.int()
    cand["loc"][0, :m, 0] = torch.arange(m).float()
    cand = {k: v.to(dev) for k, v in cand.items()}
    det = ops.nms_topk(cand, 0.6, 50)
    torch.cuda.synchronize()
    keep = O.batched_nms(box, score, cls.long(), 0.6)[:50]
    k = int(det["counts"][0])
    assert k == keep.shape[0]
    assert torch.equal(det["idx"][0, :k].cpu().long(), keep)          # bit-exact kept indices
    assert torch.equal(det["box"][0, :k].cpu(), box[keep])


def test_nms_per_class_branch_over_40000(dev):
    """>= 40000 candidates: detectron2's batched_nms runs nms per class without the coordinate trick."""
    n, m = 1, 41000
    g = torch.Generator().manual_seed(9)
    xy = torch.rand((m, 2), generator=g) * 900
    wh = torch.rand((m, 2), generator=g) * 120 + 4
    box = torch.cat([xy - 50, xy - 50 + wh], 1)
    cls = torch.randint(0, 80, (m,), generator=g)
    score = torch.rand((m,), generator=g)
    cap = 65536
    cand = dict(box=torch.zeros((n, cap, 4)), score=torch.zeros((n, cap)), cls=torch.zeros((n, cap), dtype=torch.int32),
                loc=torch.zeros((n, cap, 2)), counts=torch.tensor([m], dtype=torch.int32))
    cand["box"][0, :m], cand["score"][0, :m], cand["cls"][0, :m] = box, score, cls.int()
    cand = {k: v.to(dev) for k, v in cand.items()}
    det = ops.nms_topk(cand, 0.6, 50)
    torch.cuda.synchronize()
    keep = O.batched_nms(box, score, cls.long(), 0.6)[:50]
    assert torch.equal(det["idx"][0, :int(det["counts"][0])].cpu().long(), keep)


def test_nms_prefix_sort_falls_back_to_the_full_sort(dev):
    """More than 2 x NEED candidates: the kernel first sorts only a prefix of the score order (the bins holding the ~2048 best).  Here the
    6000 best-scoring candidates are one and the same box, so that prefix yields a single survivor and the kernel has to redo the job
    on all candidates; the result must still be detectron2's batched_nms."""
    m, cap = 12000, 16384
    g = torch.Generator().manual_seed(21)
    i = torch.arange(m)
    box = torch.stack([(i % 150) * 12.0, (i // 150) * 12.0, (i % 150) * 12.0 + 8, (i // 150) * 12.0 + 8], 1)
    score = torch.rand((m,), generator=g) * 0.4
    box[:6000] = torch.tensor([3000.0, 3000.0, 3100.0, 3100.0])
    score[:6000] = 0.5 + torch.rand((6000,), generator=g) * 0.4
    cls = torch.zeros((m,), dtype=torch.long)
    cand = dict(box=torch.zeros((1, cap, 4)), score=torch.zeros((1, cap)), cls=torch.zeros((1, cap), dtype=torch.int32),
                loc=torch.zeros((1, cap, 2)), counts=torch.tensor([m], dtype=torch.int32))
    cand["box"][0, :m], cand["score"][0, :m] = box, score
    cand = {k: v.to(dev) for k, v in cand.items()}
    det = ops.nms_topk(cand, 0.6, 50)
    torch.cuda.synchronize()
    keep = O.batched_nms(box, score, cls, 0.6)[:50]
    assert int(det["counts"][0]) == 50
    assert torch.equal(det["idx"][0].cpu().long(), keep)


def test_sort_is_stable_descending_full_order(dev):
    """topk=64 on non-overlapping boxes returns the 64 best by (score desc, index asc): checks the radix sort."""
    m, cap = 20000, 32768
    g = torch.Generator().manual_seed(11)
    score = (torch.randint(0, 5000, (m,), generator=g).float() + 1) / 5001.0      # many exact ties
    i = torch.arange(m)
    box = torch.stack([(i % 200) * 10.0, (i // 200) * 10.0, (i % 200) * 10.0 + 5, (i // 200) * 10.0 + 5], 1)
    cand = dict(box=torch.zeros((1, cap, 4)), score=torch.zeros((1, cap)), cls=torch.zeros((1, cap), dtype=torch.int32),
                loc=torch.zeros((1, cap, 2)), counts=torch.tensor([m], dtype=torch.int32))
    cand["box"][0, :m], cand["score"][0, :m] = box, score
    cand = {k: v.to(dev) for k, v in cand.items()}
    det = ops.nms_topk(cand, 0.6, 64)
    torch.cuda.synchronize()
    order = torch.sort(score, descending=True, stable=True)[1][:64]
    assert torch.equal(det["idx"][0].cpu().long(), order)


def test_roi_heads_match_reference_fixture(dev):
    """Crafted boxes (level boundaries, zero-area, out-of-image, huge): levels exact, pooled features, selected mask,
    mask-IoU against what the reference's ROIPooler / SpatialAttentionMaskHead / MaskIoUHead produced."""
    from .helpers import build_gpu_model
    g, bb = golden("roi_crafted"), golden("backbone_small")
    model, _ = build_gpu_model()
    feats = {k: bb[k].to(dev).contiguous(memory_format=torch.channels_last) for k in ("p3", "p4", "p5")}
    k = 7
    det = dict(box=torch.zeros((2, k, 4)), score=torch.zeros((2, k)), cls=torch.zeros((2, k), dtype=torch.int64),
               loc=torch.zeros((2, k, 2)), counts=torch.zeros((2,), dtype=torch.int32))
    for i in range(2):
        r = g["img{}".format(i)]
        m = r["boxes"].shape[0]
        det["box"][i, :m], det["score"][i, :m], det["cls"][i, :m], det["counts"][i] = r["boxes"], r["scores"], r["classes"], m
    det = {kk: v.to(dev) for kk, v in det.items()}
    out = model.roi_heads.forward_padded(feats, det, [(64, 96), (64, 96)], want=("roi_feat", "levels", "mask_logits", "maskiou"))
    torch.cuda.synchronize()
    rows = [i * k + j for i in range(2) for j in range(int(det["counts"][i]))]
    assert torch.equal(out["levels"].cpu()[rows].long(), g["levels"])
    close(out["roi_feat"][rows].permute(0, 3, 1, 2), g["roi_feat"], 1e-5, "roi_feat")
    cls = torch.cat([g["img0"]["classes"], g["img1"]["classes"]])
    sel = g["mask_logits"][torch.arange(len(rows)), cls]
    close_abs(out["mask_logits_selected"][rows], sel, 1e-3, "selected mask logits")
    close_abs(out["maskiou"][rows], g["maskiou"], 1e-3, "maskiou")
    start = 0
    for i in range(2):
        r = g["img{}".format(i)]
        m = r["boxes"].shape[0]
        close_abs(out["pred_masks"][i, :m], r["pred_masks"], 1e-3, "pred_masks")
        close_abs(out["mask_scores"][i, :m], r["mask_scores"], 1e-3, "mask_scores")
    # all-class predictor through the reference signature
    full = model.roi_heads.mask_head(g["roi_feat"].to(dev))
    close_abs(full, g["mask_logits"], 1e-3, "mask logits, all classes")
    miou = model.roi_heads.maskiou_head(g["roi_feat"].to(dev), torch.cat([g["img0"]["pred_masks"], g["img1"]["pred_masks"]]).to(dev))
    close_abs(miou, g["maskiou"], 1e-3, "maskiou via reference signature")


def test_pooler_by_area_and_roialign_v1_match_reference(dev):
    """The pooler's other documented modes (pooler.py:121-152 level assignment by area, :243-248 ROIAlign v1 = unaligned) against
    what the reference's own ROIPooler produced on the crafted boxes (canonical size 40 spreads them over the three levels)."""
    g, bb = golden("roi_crafted"), golden("backbone_small")
    feats = [ops.as_view(bb[k].to(dev)) for k in ("p3", "p4", "p5")]
    k = 7
    box = torch.zeros((2, k, 4))
    counts = torch.zeros((2,), dtype=torch.int32)
    for i in range(2):
        b = g["img{}".format(i)]["boxes"]
        box[i, :b.shape[0]], counts[i] = b, b.shape[0]
    y = torch.empty((2 * k, 14, 14, 256), device=dev)
    levels = ops.roi_align_ratio(feats, (1 / 8, 1 / 16, 1 / 32), box.to(dev), counts.to(dev), torch.full((2,), 64.0 * 96.0, device=dev), 14, 0, y, 3,
                                 aligned=False, assign_by_area=True, canonical_box_size=40.0, canonical_level=4)
    torch.cuda.synchronize()
    rows = [i * k + j for i in range(2) for j in range(int(counts[i]))]
    assert torch.equal(levels.cpu()[rows].long(), g["levels_area"])
    close(y[rows].permute(0, 3, 1, 2), g["roi_feat_area_v1"], 1e-5, "roi_feat (area, v1)")
    # through the module: cfg POOLER_TYPE ROIAlign + ASSIGN_CRITERION area are accepted (both used to be refused)
    from centermask2_amd.modeling.centermask.center_heads import ROIPooler
    p = ROIPooler(14, (1 / 8, 1 / 16, 1 / 32), 0, "ROIAlign", canonical_box_size=40, canonical_level=4, assign_crit="area")
    assert (p.aligned, p.assign_crit) == (False, "area")
    with pytest.raises(NotImplementedError):
        ROIPooler(14, (1 / 8,), 0, "ROIPool")


def test_roi_heads_empty_image(dev):
    from .helpers import build_gpu_model
    from centermask2_amd.structures import Boxes, Instances
    bb = golden("backbone_small")
    model, _ = build_gpu_model()
    feats = {k: bb[k][:1].to(dev).contiguous(memory_format=torch.channels_last) for k in ("p3", "p4", "p5")}
    e = Instances((64, 96))
    e.pred_boxes = Boxes(torch.zeros((0, 4), device=dev))
    e.pred_classes = torch.zeros((0,), dtype=torch.int64, device=dev)
    e.scores = torch.zeros((0,), device=dev)
    res = model.roi_heads.forward_with_given_boxes(feats, [e])
    assert tuple(res[0].pred_masks.shape) == (0, 1, 28, 28) and tuple(res[0].mask_scores.shape) == (0,)
