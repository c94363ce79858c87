"""Time ONE Winograd variant (argv[1] = tune_wm) over the layer shapes; 20 warm + 20 timed launches each."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centermask2_amd import ops, _lib
from centermask2_amd.ops import View
dev = torch.device("cuda:0"); B = 8
WM = int(sys.argv[1])
SHAPES = [("stem_2", 400, 640, 64, 64), ("OSA2_x", 200, 320, 128, 128), ("OSA3_0", 100, 160, 256, 160), ("OSA3_x", 100, 160, 160, 160),
          ("OSA4_x", 50, 80, 192, 192), ("OSA5_x", 25, 40, 224, 224), ("fcos_p3", 100, 160, 256, 256), ("fcos_p4", 50, 80, 256, 256),
          ("roi", 14, 14, 256, 256)]
lib = _lib.load()
out = []
for name, h, w, cin, cout in SHAPES:
    n = 400 if name == "roi" else B
    x = View(torch.randn((n, h, w, cin), device=dev)); pc = ops.PackedConv(torch.randn((cout, cin, 3, 3)) * 0.05, None, None, dev)
    y = View(torch.empty((n, h, w, cout), device=dev))
    d = (_lib.ConvDesc * 1)(); ops._fill_desc(d[0], x, pc, y, True, None, None, False, False)
    d[0].tune_wm, d[0].tune_sc, d[0].tune_wn = WM, 16, 2
    for _ in range(20): assert lib.cmk_conv2d_nhwc(ctypes.byref(d[0]), ops._stream()) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): lib.cmk_conv2d_nhwc(ctypes.byref(d[0]), ops._stream())
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    out.append("%s %.3f/%.0f" % (name, ms, 2.0 * n * h * w * cin * cout * 9 / ms / 1e9))
print("wm%d " % WM + " ".join(out))
