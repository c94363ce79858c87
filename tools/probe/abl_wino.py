"""Time the 2-WG Winograd kernel of alternative (ablation) builds of the library; results are wrong by design.
usage: abl_wino.py <lib.so> [<lib.so> ...]   (one subprocess per library; warm clocks: 30 untimed + 30 timed launches)"""
import sys, os, subprocess
if len(sys.argv) > 2:
    for p in sys.argv[1:]:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), p], capture_output=True, text=True)
        print(r.stdout.strip().split("\n")[-1] if r.stdout.strip() else "FAILED " + p + r.stderr[-300:], flush=True)
    sys.exit(0)
import ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from centermask2_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
from centermask2_amd import ops
from centermask2_amd.ops import View
lib = _lib.load()
dev = torch.device("cuda:0")
out = []
for name, n, h, w, cin, cout in [("OSA2_x", 8, 200, 320, 128, 128), ("fcos_p3", 8, 100, 160, 256, 256), ("roi", 400, 14, 14, 256, 256)]:
    x = View(torch.randn((n, h, w, cin), device=dev)); pc = ops.PackedConv(torch.randn((cout, cin, 3, 3)) * 0.05, None, None, dev)
    y = View(torch.empty((n, h, w, cout), device=dev))
    d = (_lib.ConvDesc * 1)(); ops._fill_desc(d[0], x, pc, y, True, None, None, False, False)
    d[0].tune_wm, d[0].tune_sc, d[0].tune_wn = 6, 16, 2
    for _ in range(30): lib.cmk_conv2d_nhwc(ctypes.byref(d[0]), ops._stream())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): lib.cmk_conv2d_nhwc(ctypes.byref(d[0]), ops._stream())
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 30
    out.append("%s %.3f ms %5.1f" % (name, ms, 2.0 * n * h * w * cin * cout * 9 / ms / 1e9))
print("%-24s" % os.path.basename(sys.argv[1])[7:-3], " | ".join(out))
