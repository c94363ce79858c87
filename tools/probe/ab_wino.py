"""Same-session A/B of the Winograd kernel in two builds of the library: ab_wino.py <libA.so> <libB.so> [rounds]
(one subprocess per measurement, alternating A B A B ..., so that box-to-box and clock drift cancel)"""
import sys, os, subprocess
if len(sys.argv) >= 3 and sys.argv[1] != "--one":
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    res = {sys.argv[1]: [], sys.argv[2]: []}
    for _ in range(rounds):
        for lib in sys.argv[1:3]:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", lib], capture_output=True, text=True)
            res[lib].append([float(v) for v in r.stdout.strip().split("\n")[-1].split()])
    names = ["stem_2", "OSA2_x", "OSA3_0", "OSA3_x", "OSA4_x", "OSA5_x", "fcos_p3", "fcos_p4", "roi"]
    for lib in sys.argv[1:3]:
        best = [min(r[i] for r in res[lib]) for i in range(len(names))]
        print("%-28s" % os.path.basename(lib), " ".join("%s %.3f" % (n, b) for n, b in zip(names, best)), "| sum %.3f" % sum(best))
    sys.exit(0)
import ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from centermask2_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[2])
from centermask2_amd import ops
from centermask2_amd.ops import View
lib = _lib.load(); dev = torch.device("cuda:0"); out = []
for name, h, w, cin, cout in [("stem_2", 400, 640, 64, 64), ("OSA2_x", 200, 320, 128, 128), ("OSA3_0", 100, 160, 256, 160), ("OSA3_x", 100, 160, 160, 160),
                              ("OSA4_x", 50, 80, 192, 192), ("OSA5_x", 25, 40, 224, 224), ("fcos_p3", 100, 160, 256, 256), ("fcos_p4", 50, 80, 256, 256), ("roi", 14, 14, 256, 256)]:
    n = 400 if name == "roi" else 8
    x = View(torch.randn((n, h, w, cin), device=dev)); pc = ops.PackedConv(torch.randn((cout, cin, 3, 3)) * 0.05, None, None, dev)
    y = View(torch.empty((n, h, w, cout), device=dev))
    d = (_lib.ConvDesc * 1)(); ops._fill_desc(d[0], x, pc, y, True, None, None, False, False)
    d[0].tune_wm, d[0].tune_sc, d[0].tune_wn = 5, 16, 2
    for _ in range(20): assert lib.cmk_conv2d_nhwc(ctypes.byref(d[0]), ops._stream()) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): lib.cmk_conv2d_nhwc(ctypes.byref(d[0]), ops._stream())
    e1.record(); torch.cuda.synchronize()
    out.append(e0.elapsed_time(e1) / 20)
print(" ".join("%.4f" % v for v in out))
