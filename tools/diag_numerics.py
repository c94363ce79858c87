"""Error of the HIP path vs the oracle per stage, direct kernels vs Winograd default. usage: diag_numerics.py BODY H W"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centermask2_amd import ops, synthetic as S
from centermask2_amd.structures import FakeImageList
from oracle import centermask_oracle as O
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from tests.helpers import build_gpu_model
body, H, W = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
model, sd = build_gpu_model(body)
x = S.make_synthetic_images(1, H, W, seed0=555)
torch.set_num_threads(16)
with torch.no_grad():
    ref_res, ref = O.centermask_inference(sd, x, [(H, W)], body, return_intermediates=True)
def stats(name, got, want):
    got, want = got.float().cpu(), want.float().cpu()
    err = (got - want).abs()
    print("   %-12s max|ref| %9.3g  max err %9.3g  (rel to max %8.2e)  rms err/rms ref %8.2e" % (name, want.abs().max(), err.max(), err.max() / want.abs().max(), err.pow(2).mean().sqrt() / want.pow(2).mean().sqrt()))
for mode in ("direct", "winograd-default"):
    os.environ["CMK_DIAG"] = mode
    ops._TUNED.clear()
    if mode == "direct":
        # force direct kernels: drop the Winograd-packed weights so the library cannot choose them
        for m in model.modules():
            if hasattr(m, "_packed_cache"): m._packed_cache = None
        orig = ops.pack_wino_weight
        ops.PackedConv_disable = True
        _init = ops.PackedConv.__init__
        def patched(self, *a, **k):
            _init(self, *a, **k); self.w_wino = None
        ops.PackedConv.__init__ = patched
    else:
        ops.PackedConv.__init__ = _init
        for m in model.modules():
            if hasattr(m, "_packed_cache"): m._packed_cache = None
    print(mode)
    with torch.no_grad():
        feats = model.backbone(x.cuda())
        for k in ("p3", "p5", "p7"): stats(k, feats[k], ref["features"][k])
        lg, reg, ctr, _ = model.proposal_generator.fcos_head([feats[k] for k in ("p3", "p4", "p5", "p6", "p7")])
        stats("logits p3", lg[0], ref["logits"][0]); stats("reg p3", reg[0], ref["bbox_reg"][0])
        out = model.inference_padded(x.cuda(), [(H, W)], want=("mask_logits", "maskiou"))
        n = ref_res[0]["scores"].shape[0]
        same = torch.equal(out["cls"][0, :n].cpu(), ref_res[0]["classes"]) and torch.equal(out["loc"][0, :n].cpu(), ref_res[0]["locations"])
        print("   labels/locations equal:", same, "n", n)
        if same:
            cls = ref_res[0]["classes"]
            stats("mask logits", out["mask_logits_selected"][:n], ref["mask_logits"][torch.arange(n), cls])
            stats("pred_masks", out["pred_masks"][0, :n], ref_res[0]["pred_masks"])
            stats("mask_scores", out["mask_scores"][0, :n], ref_res[0]["mask_scores"])
