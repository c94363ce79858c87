"""3x3 stride-1 convs: the best F(4x4,3x3) Winograd form (tune 6/16 | 6/64, fp32 MFMA) against the direct form on bf16-split products
(conv_sp3.hip, tune 11 / pieces / geometry; opt-in), same session, interleaved, best of `rounds`; max distance of each from a float64 conv
of image 0.  python tools/bench_sp3.py [rounds] [pieces]"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centermask2_amd import ops, _lib
from centermask2_amd.ops import View
ops.ALLOW_SPLIT_BF16 = ops.ALLOW_SPLIT_F16 = True
dev = torch.device("cuda:0"); B = 8
SHAPES = [("stem_2", 400, 640, 64, 64), ("OSA2_x", 200, 320, 128, 128), ("OSA3_0", 100, 160, 256, 160), ("OSA3_x", 100, 160, 160, 160),
          ("OSA4_0", 50, 80, 512, 192), ("OSA4_x", 50, 80, 192, 192), ("OSA5_x", 25, 40, 224, 224),
          ("fpn_p3", 100, 160, 256, 256), ("fpn_p4", 50, 80, 256, 256), ("fpn_p5", 25, 40, 256, 256), ("roi", 14, 14, 256, 256)]
lib = _lib.load()
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
pieces = int(sys.argv[2]) if len(sys.argv) > 2 else 2
def timeit(d, it=8):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): lib.cmk_conv2d_nhwc(ctypes.byref(d), ops._stream())
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
print("%-8s %8s %8s | %8s %8s %8s %8s | %6s %8s %9s %9s" % ("layer", "w6 ms", "w6s ms", "geo0", "geo1", "geo2", "geo3", "dirTF", "wino/sp3", "err wino", "err sp3"), flush=True)
for name, h, w, cin, cout in SHAPES:
    roi = name.startswith("roi"); n = 400 if roi else B
    x = View(torch.randn((n, h, w, cin), device=dev)); wt = torch.randn((cout, cin, 3, 3)) * (2.0 / (9 * cin)) ** 0.5
    pc = ops.PackedConv(wt, torch.rand(cout) + 0.5, torch.randn(cout) * 0.1, dev)
    tvs = [(6, 16, 2 if roi else 1), (6, 64, 2 if roi else 1)] + [(11, pieces, g) for g in range(4)]
    ref = torch.nn.functional.conv2d(x.t[:1].permute(0, 3, 1, 2).double(), wt.to(dev).double(), padding=1)
    ref = (ref * pc.scale.double()[None, :, None, None] + pc.shift.double()[None, :, None, None]).relu().permute(0, 2, 3, 1)
    best, errs = [], []
    for tv in tvs:
        y = View(torch.full((n, h, w, cout), float("nan"), device=dev))
        d = (_lib.ConvDesc * 1)(); ops._fill_desc(d[0], x, pc, y, True, None, None, False, False)
        d[0].tune_wm, d[0].tune_sc, d[0].tune_wn = tv
        if lib.cmk_conv2d_nhwc(ctypes.byref(d[0]), ops._stream()) != 0:
            best.append(float("inf")); errs.append(float("nan")); continue
        torch.cuda.synchronize()
        assert bool(torch.isfinite(y.t).all()), (name, tv, "unwritten outputs")
        errs.append(float((y.t[:1].double() - ref).abs().max()))
        best.append(min(timeit(d[0]) for _ in range(rounds)))
    fl = 2.0 * n * h * w * 9 * cin * cout
    bw, bs = min(best[:2]), min(best[2:])
    print("%-8s %8.3f %8.3f | %8.3f %8.3f %8.3f %8.3f | %6.1f %8.2f %9.2e %9.2e" % (name, best[0], best[1], best[2], best[3], best[4], best[5], fl / bs / 1e9, bw / bs,
          max(e for e in errs[:2] if e == e), max(e for e in errs[2:] if e == e)), flush=True)
