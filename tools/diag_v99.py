"""V-99-eSE at 8 x 3x800x1280 (BASELINE config 5's per-GPU workload): how far apart are (a) the CPU oracle on THIS host and the fixture the
reference produced in the build container, (b) the HIP path and the oracle on this host, (c) both against a float64 run of the oracle
(the arithmetic's own ground truth).  Prints numbers; the test tolerances in tests/test_gpu_model.py are set from them."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centermask2_amd import ops, synthetic as S
from oracle import centermask_oracle as O
from tests.helpers import build_gpu_model, golden, match_detections
torch.set_num_threads(min(16, os.cpu_count() or 1))
dev = torch.device("cuda:0")
g = golden("e2e_v99_800x1280")
model, sd = build_gpu_model("V-99-eSE")
ops.load_tuned(os.path.join(os.path.dirname(ops.__file__), "tuned", "mi355x_V-99-eSE_b8_800x1280.json"))
x = S.make_synthetic_images(8, 800, 1280, seed0=1234)
sizes = [(800, 1280)] * 8
with torch.no_grad():
    out = model.inference_padded(x.to(dev), sizes)
    feats = model.backbone(x.to(dev))
    lg, reg, ctr, _ = model.proposal_generator.fcos_head([feats[k] for k in ("p3", "p4", "p5", "p6", "p7")])
torch.cuda.synchronize()
res = model.results_from_padded(out, sizes)
t0 = time.time()
want, wi = O.centermask_inference(sd, x[:1], sizes[:1], "V-99-eSE", return_intermediates=True)
print("oracle fp32: %.1f s" % (time.time() - t0))
sd64 = {k: v.double() for k, v in sd.items()}
t0 = time.time()
with torch.no_grad():          # float64 up to the proposals (the ROI kernels of the oracle are float32 C code)
    f64 = O.backbone_forward(sd64, x[:1].double(), "V-99-eSE")
    l64, r64, c64 = O.fcos_head_forward(sd64, [f64[k] for k in ("p3", "p4", "p5", "p6", "p7")])
    p64 = O.fcos_predict_proposals([t.float() for t in l64], [t.float() for t in r64], [t.float() for t in c64])
w64, i64 = p64, dict(features=f64, logits=l64, bbox_reg=r64, ctrness=c64)
print("oracle fp64: %.1f s" % (time.time() - t0))
def rel(a, b): return float((a.double() - b.double()).norm() / b.double().norm())
def mx(a, b): return float((a.double() - b.double()).abs().max())
for k in ("p3", "p5", "p7"):
    print("feature %s: oracle32 vs 64 rel %.2e max %.2e | HIP vs 64 rel %.2e max %.2e | HIP vs oracle32 max %.2e (absmax %.1f)" % (
        k, rel(wi["features"][k], i64["features"][k]), mx(wi["features"][k], i64["features"][k]), rel(feats[k][:1].cpu(), i64["features"][k]),
        mx(feats[k][:1].cpu(), i64["features"][k]), mx(feats[k][:1].cpu(), wi["features"][k]), float(i64["features"][k].abs().max())))
for nm, mine in (("logits", lg), ("bbox_reg", reg), ("ctrness", ctr)):
    print("%-8s: oracle32 vs 64 max %.2e | HIP vs 64 max %.2e | HIP vs oracle32 max %.2e" % (
        nm, max(mx(a, b) for a, b in zip(wi[nm], i64[nm])), max(mx(a[:1].cpu(), b) for a, b in zip(mine, i64[nm])), max(mx(a[:1].cpu(), b) for a, b in zip(mine, wi[nm]))))
r, w, w6, h = g["img0"], want[0], w64[0], res[0]
for name, a, b in (("host oracle32 vs fixture", w, r), ("host oracle32 vs oracle64", w, w6), ("fixture vs oracle64", r, w6)):
    try:
        p = match_detections(a["scores"], a["classes"], a["locations"], b["scores"].float(), b["classes"], b["locations"].float(), tol=1e-3)
        print("%s: same detections; boxes max %.3e px, scores max %.2e" % (name, mx(a["boxes"], b["boxes"][p]), mx(a["scores"], b["scores"][p])))
    except AssertionError as e:
        print(name, "DIFFERENT:", str(e)[:200])
for name, b in (("oracle32", w), ("oracle64", w6), ("fixture", r)):
    try:
        p = match_detections(h.scores, h.pred_classes, h.locations, b["scores"].float(), b["classes"], b["locations"].float(), tol=1e-3)
        print("HIP vs %s: same detections; boxes max %.3e px, scores max %.2e, order identical %s" % (
            name, mx(h.pred_boxes.tensor.cpu(), b["boxes"][p]), mx(h.scores.cpu(), b["scores"][p]), bool((p == torch.arange(len(p))).all())))
    except AssertionError as e:
        print("HIP vs", name, "DIFFERENT:", str(e)[:200])
