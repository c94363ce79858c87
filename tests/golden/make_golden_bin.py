"""The six `.bin` files of the reference's deployment flow for one image, produced with the REFERENCE's own tuple
helper and read back with the reference's own reader steps.

    python tests/golden/make_golden_bin.py        # needs /root/reference; writes tests/golden/bin/000000000139_{1..6}.bin + bin_expected.pt

single_flatten_to_tuple (deploy_utils.py:117-126) fixes the order (locations, mask_scores, pred_boxes, pred_classes, pred_masks, scores);
postprocess_bin_outputs.py:36-45 fixes dtypes and shapes (np.fromfile(dtype) + reshape).  The Ascend tool that writes the files is not
in the reference tree; a raw little-endian dump of each tensor in those dtypes is what its reader consumes.  Five detections are kept
(the masks are 3 KB each).  Data only.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402  (installs the d2 stand-ins and imports the reference package)

sys.path.insert(0, "/root/reference")
import deploy_utils as RD  # noqa: E402  the reference's own helpers (its third-party imports resolve to the stand-ins)

S, O = G.S, G.O
LST_DTYPE = ['float32', 'float32', 'float32', 'int64', 'float32', 'float32']          # postprocess_bin_outputs.py:37
LST_SHAPE = [(-1, 2), (-1), (-1, 4), (-1), (-1, 1, 28, 28), (-1)]                      # :38


def main():
    _, backbone, fcos, roi_heads = G.build_reference("V-39-eSE")
    sd = S.make_synthetic_state_dict("V-39-eSE", seed=0)
    G.load_synthetic(backbone, fcos, roi_heads, sd)
    x = S.make_synthetic_images(1, 256, 320, seed0=4321)
    images = G.FakeImageList(x, [(256, 320)])
    with torch.no_grad():
        feats = backbone(x)
        props, _ = G.quiet(fcos, images, feats, None)
        results, _ = G.quiet(roi_heads, images, feats, props, None)
    tup = RD.single_flatten_to_tuple(results[0][:5])                 # the reference's field order
    os.makedirs(os.path.join(HERE, "bin"), exist_ok=True)
    prefix = os.path.join(HERE, "bin", "000000000139")
    for i, (t, dt) in enumerate(zip(tup, LST_DTYPE)):
        RD.to_numpy(t).astype(dt).tofile("{}_{}.bin".format(prefix, i + 1))
    # the reference's reader (postprocess_bin_outputs.py:39-45): np.fromfile + reshape, then single_wrap_outputs
    lst = [torch.from_numpy(np.fromfile("{}_{}.bin".format(prefix, i + 1), dtype=LST_DTYPE[i])).reshape(LST_SHAPE[i]) for i in range(6)]
    inst = RD.single_wrap_outputs(lst, 256, 320)[0]
    exp = dict(locations=inst.locations, mask_scores=inst.mask_scores, pred_boxes=inst.pred_boxes.tensor, pred_classes=inst.pred_classes,
               pred_masks=inst.pred_masks, scores=inst.scores)
    for k, t in zip(("locations", "mask_scores", "pred_boxes", "pred_classes", "pred_masks", "scores"), tup):
        assert torch.equal(exp[k], t), k
    assert exp["pred_classes"].dtype == torch.int64 and tuple(exp["pred_masks"].shape) == (5, 1, 28, 28)
    torch.save({k: v.clone() for k, v in exp.items()}, os.path.join(HERE, "bin_expected.pt"))
    print("wrote", prefix + "_{1..6}.bin", [os.path.getsize("{}_{}.bin".format(prefix, i + 1)) for i in range(6)])


if __name__ == "__main__":
    main()
