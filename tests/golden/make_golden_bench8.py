"""Golden vectors for the configuration the metric is quoted on (BASELINE config 4): the REFERENCE's own modules
(VoVNet V-39-eSE + FPN, FCOS, CenterROIHeads) on ALL EIGHT images of bench.py's batch (seeds 1234..1241, 3x800x1280).

    python tests/golden/make_golden_bench8.py     # needs /root/reference; writes tests/golden/e2e_bench8_800x1280.pt

Call order of tester.py:94-104 (backbone -> proposal_generator -> roi_heads), one image at a time: nothing in inference couples
two images, and the first two images are asserted equal to the batch-2 fixture e2e_800x1280.pt.  Per image the
fixture holds the final tuple (boxes, scores, classes, locations, pred_masks, mask_scores), the candidate count and sparse
probes of p3..p7 and of the FCOS logits.  The oracle must agree with the reference before anything is written.  Data only.
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402  (installs the d2 stand-ins and imports the reference package)

S, O = G.S, G.O
BODY = "V-39-eSE"
SEED0, B = 1234, 8


def main():
    _, backbone, fcos, roi_heads = G.build_reference(BODY)
    sd = S.make_synthetic_state_dict(BODY, seed=0)
    G.load_synthetic(backbone, fcos, roi_heads, sd)
    x = S.make_synthetic_images(B, 800, 1280, seed0=SEED0)
    sizes = [(800, 1280)]
    out = dict(image_seed0=torch.tensor(SEED0), weight_seed=torch.tensor(0), num_images=torch.tensor(B))
    prev = torch.load(os.path.join(HERE, "e2e_800x1280.pt"), weights_only=True)
    for i in range(B):
        xi = x[i:i + 1]
        with torch.no_grad():
            feats = backbone(xi)
            props, _ = G.quiet(fcos, G.FakeImageList(xi, sizes), feats, None)
            results, _ = G.quiet(roi_heads, G.FakeImageList(xi, sizes), feats, props, None)
            rl, rr, rc, _ = fcos.fcos_head([feats[k] for k in ("p3", "p4", "p5", "p6", "p7")])
        ores, inter = O.centermask_inference(sd, xi, sizes, BODY, return_intermediates=True)
        r, o = G.inst_to_dict(results[0]), ores[0]
        assert torch.equal(r["classes"], o["classes"]) and torch.equal(r["locations"], o["locations"])
        G.close(o["boxes"], r["boxes"], 1e-6, "boxes")
        G.close(o["scores"], r["scores"], 1e-6, "scores")
        G.close(o["pred_masks"], r["pred_masks"], 1e-5, "masks")
        G.close(o["mask_scores"], r["mask_scores"], 1e-5, "mask_scores")
        if i < 2:       # the batch-2 fixture made from the same modules: batching changes nothing that is compared exactly
            p = prev["img{}".format(i)]
            assert torch.equal(p["classes"], r["classes"]) and torch.equal(p["locations"], r["locations"])
            G.close(r["boxes"], p["boxes"], 1e-6, "boxes vs batch-2 fixture")
            G.close(r["pred_masks"], p["pred_masks"], 1e-5, "masks vs batch-2 fixture")
        ncand = inter["candidates"][0]["scores"].shape[0]
        sc = r["scores"]
        gaps = (sc[:-1] - sc[1:]) if sc.numel() > 1 else torch.ones(1)
        lv = O.assign_boxes_to_levels_by_ratio(r["boxes"], torch.full((r["boxes"].shape[0],), 800.0 * 1280.0))
        print("bench8 img", i, "cands", ncand, "dets", sc.shape[0], "levels", torch.bincount(lv, minlength=3).tolist(),
              "smallest score gap {:.3e}".format(float(gaps.min())))
        d = dict(**{k: v.clone() for k, v in r.items()}, num_candidates=torch.tensor(ncand))
        for k in ("p3", "p4", "p5", "p6", "p7"):
            d[k] = G.probe(feats[k], n=256)
        for l in range(5):
            d["logits{}".format(l)] = G.probe(rl[l], n=256)
            d["reg{}".format(l)] = G.probe(rr[l], n=256)
            d["ctr{}".format(l)] = G.probe(rc[l], n=256)
        out["img{}".format(i)] = d
    path = os.path.join(HERE, "e2e_bench8_800x1280.pt")
    torch.save(out, path)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
