"""F(2x2,3x3) (tune_wm 5) vs F(4x4,3x3) (tune_wm 6) per 3x3 layer shape of the model, same session, interleaved; also checks that the
two agree.  python tools/bench_wino6.py [rounds]"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centermask2_amd import ops, _lib
from centermask2_amd.ops import View
dev = torch.device("cuda:0"); B = 8
SHAPES = [("stem_2", 400, 640, 64, 64), ("OSA2_x", 200, 320, 128, 128), ("OSA3_0", 100, 160, 256, 160), ("OSA3_x", 100, 160, 160, 160),
          ("OSA4_0", 50, 80, 512, 192), ("OSA4_x", 50, 80, 192, 192), ("OSA5_0", 25, 40, 768, 224), ("OSA5_x", 25, 40, 224, 224),
          ("fpn_p3", 100, 160, 256, 256), ("fcos_p4", 50, 80, 256, 256), ("fcos_p5", 25, 40, 256, 256), ("roi", 14, 14, 256, 256)]
lib = _lib.load()
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
def timeit(d, it=6):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): lib.cmk_conv2d_nhwc(ctypes.byref(d), ops._stream())
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
print("%-8s %9s %8s %9s %8s %7s %10s" % ("layer", "F2 ms", "algTF", "F4 ms", "algTF", "F2/F4", "max|diff|"), flush=True)
tot = [0.0, 0.0]
for name, h, w, cin, cout in SHAPES:
    n = 400 if name == "roi" else B
    x = View(torch.randn((n, h, w, cin), device=dev)); pc = ops.PackedConv(torch.randn((cout, cin, 3, 3)) * (2.0 / (9 * cin)) ** 0.5, None, None, dev)
    ys = [View(torch.empty((n, h, w, cout), device=dev)) for _ in range(2)]
    ds = []
    for k, tv in enumerate(((5, 16, 2), (6, 16, 2) if name == "roi" else (6, 16, 1))):
        d = (_lib.ConvDesc * 1)(); ops._fill_desc(d[0], x, pc, ys[k], True, None, None, False, False)
        d[0].tune_wm, d[0].tune_sc, d[0].tune_wn = tv
        assert lib.cmk_conv2d_nhwc(ctypes.byref(d[0]), ops._stream()) == 0, lib.cmk_last_error()
        ds.append(d)
    torch.cuda.synchronize()
    diff = float((ys[0].t - ys[1].t).abs().max()); mx = float(ys[0].t.abs().max())
    best = [1e9, 1e9]
    for _ in range(rounds):
        for k in range(2): best[k] = min(best[k], timeit(ds[k][0]))
    fl = 2.0 * n * h * w * cin * cout * 9
    tot[0] += best[0]; tot[1] += best[1]
    print("%-8s %9.3f %8.1f %9.3f %8.1f %7.2f %10.2e (max %.1f)" % (name, best[0], fl / best[0] / 1e9, best[1], fl / best[1] / 1e9, best[0] / best[1], diff, mx), flush=True)
print("sum F2 %.3f ms, F4 %.3f ms" % tuple(tot))
