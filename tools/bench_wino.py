"""Direct (best tuned variant) vs Winograd per 3x3 layer shape."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centermask2_amd import ops, _lib
from centermask2_amd.ops import View
dev = torch.device("cuda:0"); B = 8
SHAPES = [("stem_2", 400, 640, 64, 64), ("OSA2_x", 200, 320, 128, 128), ("OSA3_0", 100, 160, 256, 160), ("OSA3_x", 100, 160, 160, 160),
          ("OSA4_x", 50, 80, 192, 192), ("OSA5_x", 25, 40, 224, 224), ("fcos_p3", 100, 160, 256, 256), ("fcos_p4", 50, 80, 256, 256),
          ("roi", 14, 14, 256, 256)]
lib = _lib.load()
def timeit(d, it=4):
    lib.cmk_conv2d_nhwc(ctypes.byref(d), ops._stream())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): rc = lib.cmk_conv2d_nhwc(ctypes.byref(d), ops._stream())
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
print("%-8s %9s %9s %9s %9s" % ("layer", "direct ms", "TF", "wino ms", "algTF"))
for name, h, w, cin, cout in SHAPES:
    n = 400 if name == "roi" else B
    x = View(torch.randn((n, h, w, cin), device=dev)); pc = ops.PackedConv(torch.randn((cout, cin, 3, 3)) * 0.05, None, None, dev)
    y = View(torch.empty((n, h, w, cout), device=dev))
    d = (_lib.ConvDesc * 1)(); ops._fill_desc(d[0], x, pc, y, True, None, None, False, False)
    best = 1e9
    for wn in range(1, 8):
        for wm in (1, 2):
            for sc in (16, 32):
                d[0].tune_wm, d[0].tune_sc, d[0].tune_wn = wm, sc, wn
                if lib.cmk_conv2d_nhwc(ctypes.byref(d[0]), ops._stream()) == 0: best = min(best, timeit(d[0]))
    d[0].tune_wm, d[0].tune_sc, d[0].tune_wn = 5, 16, 2
    tw4r = timeit(d[0])
    fl = 2.0 * n * h * w * cin * cout * 9
    print("%-8s %9.3f %9.1f %9.3f %9.1f" % (name, best, fl / best / 1e9, tw4r, fl / tw4r / 1e9))
