"""BASELINE configs 2-4 side by side on one GPU: VoVNet backbone only, backbone+FPN+FCOS (no mask branch), full model.
usage: bench_stages.py [body] [batch]   (eager launches timed with events over 10 iterations after 3 warm-ups)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centermask2_amd import ops, synthetic as S
from centermask2_amd.config import get_cfg, config_path
from centermask2_amd.modeling import build_model

body = sys.argv[1] if len(sys.argv) > 1 else "V-39-eSE"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
cfg = get_cfg()
cfg.merge_from_file(config_path("centermask_V_99_eSE_FPN_ms_3x.yaml" if body == "V-99-eSE" else "centermask_V_39_eSE_FPN_ms_3x.yaml"))
cfg.merge_from_list(["MODEL.DEVICE", "cuda", "MODEL.VOVNET.CONV_BODY", body])
cfg.freeze()
model = build_model(cfg).eval()
model.load_state_dict(S.make_synthetic_state_dict(body, 0))
table = os.path.join(os.path.dirname(ops.__file__), "tuned", "mi355x_{}_b{}_800x1280.json".format(body, B))
if os.path.exists(table):
    ops.load_tuned(table)
x = S.make_synthetic_images(B, 800, 1280, seed0=1234).cuda()
sizes = [(800, 1280)] * B


def timed(fn, it=10, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it


with torch.no_grad():
    t_bu = timed(lambda: model.backbone.bottom_up.forward_views(x))
    t_bb = timed(lambda: model.backbone(x))

    def det():
        f = model.backbone(x)
        return model.proposal_generator.forward_padded(f)
    t_det = timed(det)
    t_full = timed(lambda: model.inference_padded(x, sizes))
print("{} bs={} 3x800x1280 fp32, ms per batch (img/s):".format(body, B))
for name, t in (("VoVNet bottom-up (BASELINE config 2)", t_bu), ("+ FPN", t_bb), ("+ FCOS head, decode, NMS (config 3)", t_det), ("full model (config 4)", t_full)):
    print("  %-40s %8.2f  (%.1f)" % (name, t, 1e3 * B / t))
