"""Golden vectors for BASELINE config 5's body: the REFERENCE's own modules (VoVNet V-99-eSE + FPN, FCOS, CenterROIHeads) end to end on one
seeded 800x1280 image.

    python tests/golden/make_golden_v99.py        # needs /root/reference; writes tests/golden/e2e_v99_800x1280.pt

The reference publishes no V-99 yaml, only the stage spec (vovnet.py:90-98); the model is the V-39 recipe with
MODEL.VOVNET.CONV_BODY = "V-99-eSE".  The oracle must agree with the reference before anything is written.  Data only.
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402  (installs the d2 stand-ins and imports the reference package)

S, O = G.S, G.O
BODY = "V-99-eSE"


def main():
    _, backbone, fcos, roi_heads = G.build_reference(BODY)
    sd = S.make_synthetic_state_dict(BODY, seed=0)
    G.load_synthetic(backbone, fcos, roi_heads, sd)
    x = S.make_synthetic_images(1, 800, 1280, seed0=1234)
    sizes = [(800, 1280)]
    with torch.no_grad():
        feats = backbone(x)
        props, _ = G.quiet(fcos, G.FakeImageList(x, sizes), feats, None)
        results, _ = G.quiet(roi_heads, G.FakeImageList(x, sizes), feats, props, None)
        rl, rr, rc, _ = fcos.fcos_head([feats[k] for k in ("p3", "p4", "p5", "p6", "p7")])
    ores, inter = O.centermask_inference(sd, x, sizes, BODY, return_intermediates=True)
    out = dict(image_seed0=torch.tensor(1234), weight_seed=torch.tensor(0))
    for k in ("p3", "p4", "p5", "p6", "p7"):
        G.close(inter["features"][k], feats[k], 1e-5, "v99 " + k)
        out[k] = G.probe(feats[k])
    for l in range(5):
        G.close(inter["logits"][l], rl[l], 1e-5, "v99 logits")
        out["logits{}".format(l)] = G.probe(rl[l])
        out["reg{}".format(l)] = G.probe(rr[l])
        out["ctr{}".format(l)] = G.probe(rc[l])
    r, o = G.inst_to_dict(results[0]), ores[0]
    assert torch.equal(r["classes"], o["classes"]) and torch.equal(r["locations"], o["locations"])
    G.close(o["boxes"], r["boxes"], 1e-6, "v99 boxes")
    G.close(o["scores"], r["scores"], 1e-6, "v99 scores")
    G.close(o["pred_masks"], r["pred_masks"], 1e-5, "v99 masks")
    G.close(o["mask_scores"], r["mask_scores"], 1e-5, "v99 mask_scores")
    lv = O.assign_boxes_to_levels_by_ratio(r["boxes"], torch.full((r["boxes"].shape[0],), 800.0 * 1280.0))
    print("v99 e2e: cands", inter["candidates"][0]["scores"].shape[0], "dets", r["scores"].shape[0], "levels", torch.bincount(lv, minlength=3).tolist())
    out["img0"] = dict(**{k: v.clone() for k, v in r.items()}, num_candidates=torch.tensor(inter["candidates"][0]["scores"].shape[0]))
    path = os.path.join(HERE, "e2e_v99_800x1280.pt")
    torch.save(out, path)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
