"""Where the end-to-end error of the opt-in direct-split table comes from: the eight bench images against the reference's fixture
(tests/golden/e2e_bench8_800x1280.pt) with the shipped table, the split3 table, and the split3 table with groups of its tune-11 entries
reverted to the shipped choice.  python tools/diag_sp3_e2e.py"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centermask2_amd import ops, synthetic as S
from tests.helpers import build_gpu_model, golden
ops.ALLOW_SPLIT_BF16 = ops.ALLOW_SPLIT_F16 = True
dev = torch.device("cuda:0")
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
base = json.load(open(os.path.join(root, "centermask2_amd/tuned/mi355x_V-39-eSE_b8_800x1280.json")))
sp3 = json.load(open(os.path.join(root, "centermask2_amd/tuned/mi355x_V-39-eSE_b8_800x1280_split3.json")))
model = build_gpu_model()[0]
g = golden("e2e_bench8_800x1280"); B = int(g["num_images"])
x = S.make_synthetic_images(B, 800, 1280, seed0=int(g["image_seed0"])).to(dev); sizes = [(800, 1280)] * B
def is_tower(k): return "+8x50x80+" in k
def is_roi(k): return "400x14x14" in k
def is_body(k): return k.startswith("k3s1") and not is_tower(k) and not is_roi(k) and "xcs256_ycs256" not in k
def is_fpn(k): return k.startswith("k3s1") and "xcs256_ycs256" in k and not is_tower(k) and not is_roi(k)
configs = [("shipped (fp32 MFMA)", base), ("six-product 1x1 only", {k: (v if v[0] != 11 else base[k]) for k, v in sp3.items()}), ("split3 (all)", sp3)]
for name, pred in (("split3, towers reverted", is_tower), ("split3, body reverted", is_body), ("split3, fpn reverted", is_fpn), ("split3, roi reverted", is_roi),
                   ("split3, only towers", lambda k: not is_tower(k)), ("split3, only body", lambda k: not is_body(k))):
    configs.append((name, {k: (base[k] if (v[0] == 11 and pred(k)) else v) for k, v in sp3.items()}))
def probe_err(t, p):
    flat = t.contiguous().reshape(-1).cpu()
    return float((flat[p["idx"]] - p["val"]).abs().max())
print("%-28s %9s %9s %9s %9s %9s %7s" % ("table", "p3", "logits", "reg", "scores", "masks", "order"))
for name, table in configs:
    ops._TUNED.clear()
    for k, v in table.items():
        if ops._variant_on_menu(v): ops._TUNED[ops._str_to_key(k)] = tuple(v)
    with torch.no_grad():
        out = model.inference_padded(x, sizes)
        feats = model.backbone(x)
        lg, reg, ctr, _ = model.proposal_generator.fcos_head([feats[k] for k in ("p3", "p4", "p5", "p6", "p7")])
        torch.cuda.synchronize()
        res = model.results_from_padded(out, sizes)
    e = {"p3": 0.0, "logits": 0.0, "reg": 0.0, "scores": 0.0, "masks": 0.0}; same = 0
    for i in range(B):
        r = g["img{}".format(i)]
        e["p3"] = max(e["p3"], max(probe_err(feats[k][i:i + 1], r[k]) for k in ("p3", "p4", "p5", "p6", "p7")))
        for l in range(5):
            e["logits"] = max(e["logits"], probe_err(lg[l][i:i + 1], r["logits{}".format(l)]))
            e["reg"] = max(e["reg"], probe_err(reg[l][i:i + 1], r["reg{}".format(l)]))
        ok = len(res[i]) == r["scores"].shape[0] and bool(torch.equal(res[i].pred_classes.cpu(), r["classes"])) and bool(torch.equal(res[i].locations.cpu(), r["locations"]))
        same += int(ok)
        if ok:
            e["scores"] = max(e["scores"], float((res[i].scores.cpu() - r["scores"]).abs().max()))
            e["masks"] = max(e["masks"], float((res[i].pred_masks.cpu() - r["pred_masks"]).abs().max()))
    print("%-28s %9.2e %9.2e %9.2e %9.2e %9.2e %5d/8" % (name, e["p3"], e["logits"], e["reg"], e["scores"], e["masks"], same), flush=True)
