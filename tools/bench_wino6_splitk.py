"""F(4x4,3x3) with the chunk loop split over 1 / 2 / 4 workgroups (tune 6/16/1/sk) and the shared-V form, on the single-round shapes of
stages 4 and 5.  python tools/bench_wino6_splitk.py [rounds]"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centermask2_amd import ops, _lib
from centermask2_amd.ops import View
dev = torch.device("cuda:0"); B = 8
SHAPES = [("OSA4_0", 50, 80, 512, 192), ("OSA4_x", 50, 80, 192, 192), ("OSA4_2_0", 50, 80, 768, 192), ("OSA5_0", 25, 40, 768, 224), ("OSA5_x", 25, 40, 224, 224),
          ("OSA5_2_0", 25, 40, 1024, 224), ("fpn_p4", 50, 80, 256, 256), ("fpn_p5", 25, 40, 256, 256), ("OSA3_0", 100, 160, 256, 160)]
lib = _lib.load()
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
VARS = [(6, 16, 1), (6, 16, 1, 2), (6, 16, 1, 4), (6, 64, 1)]
def timeit(d, it=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): lib.cmk_conv2d_nhwc(ctypes.byref(d), ops._stream())
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
print("%-9s " % "layer" + " ".join("%12s" % str(v) for v in VARS), flush=True)
for name, h, w, cin, cout in SHAPES:
    x = View(torch.randn((B, h, w, cin), device=dev)); pc = ops.PackedConv(torch.randn((cout, cin, 3, 3)) * (2.0 / (9 * cin)) ** 0.5, None, None, dev)
    ds, keep = [], []
    for tv in VARS:
        y = View(torch.empty((B, h, w, cout), device=dev))
        d = (_lib.ConvDesc * 1)(); ops._fill_desc(d[0], x, pc, y, True, None, None, False, False)
        keep.append((y, ops._set_variant(d, 1, tv)))
        ok = lib.cmk_conv2d_nhwc(ctypes.byref(d[0]), ops._stream()) == 0
        ds.append(d if ok else None)
    torch.cuda.synchronize()
    best = [1e9] * len(VARS)
    for _ in range(rounds):
        for k, d in enumerate(ds):
            if d is not None: best[k] = min(best[k], timeit(d[0]))
    err = [float((keep[k][0].t - keep[0][0].t).abs().max()) if ds[k] is not None else float("nan") for k in range(len(VARS))]
    print("%-9s " % name + " ".join("%12s" % ("%.3f" % b if b < 1e8 else "n/a") for b in best) + "   max|diff| vs unsplit " + " ".join("%.1e" % e for e in err[1:]), flush=True)
