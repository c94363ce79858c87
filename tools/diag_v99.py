import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centermask2_amd import ops, synthetic as S
from oracle import centermask_oracle as O
from tests.helpers import build_gpu_model
model, sd = build_gpu_model("V-99-eSE")
x = S.make_synthetic_images(1, 128, 192, seed0=555)
torch.set_num_threads(16)
with torch.no_grad():
    ref_res, ref = O.centermask_inference(sd, x, [(128, 192)], "V-99-eSE", return_intermediates=True)
    out = model.inference_padded(x.cuda(), [(128, 192)], want=("roi_feat", "levels", "mask_logits", "maskiou"))
n = 50
print("levels equal", torch.equal(out["levels"][:n].cpu().long(), ref["levels"]))
box_err = (out["box"][0, :n].cpu() - ref_res[0]["boxes"]).abs().max(dim=1)[0]
print("box err max", float(box_err.max()))
rf = out["roi_feat"][:n].permute(0, 3, 1, 2).cpu(); e = (rf - ref["roi_feat"]).abs().flatten(1).max(dim=1)[0]
print("roi_feat err per roi (top5)", torch.topk(e, 5))
cls = ref_res[0]["classes"]
ml = (out["mask_logits_selected"][:n].cpu() - ref["mask_logits"][torch.arange(n), cls]).abs().flatten(1).max(dim=1)[0]
print("mask logit err per roi (top5)", torch.topk(ml, 5))
for k in ("p3", "p4", "p5"):
    f = model.backbone(x.cuda())[k].cpu(); r = ref["features"][k]
    err = (f - r).abs()
    print(k, "max err", float(err.max()), "at", tuple(int(v) for v in torch.nonzero(err == err.max())[0]), "ref there", float(r[tuple(torch.nonzero(err == err.max())[0])]))
