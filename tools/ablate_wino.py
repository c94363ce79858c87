"""Ablation of conv_wino4s_kernel (guide rule 17): time builds with one cost removed. Outputs are wrong by construction."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centermask2_amd import ops, _lib
from centermask2_amd.ops import View
lib = _lib.load()
raw = ctypes.CDLL(_lib.LIB_PATH)
dev = torch.device("cuda:0")
def timeit(d, it=5):
    lib.cmk_conv2d_nhwc(ctypes.byref(d), ops._stream())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): lib.cmk_conv2d_nhwc(ctypes.byref(d), ops._stream())
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
for name, h, w, cin, cout in [("OSA2_x", 200, 320, 128, 128), ("fcos_p3", 100, 160, 256, 256)]:
    x = View(torch.randn((8, h, w, cin), device=dev)); pc = ops.PackedConv(torch.randn((cout, cin, 3, 3)) * 0.05, None, None, dev)
    y = View(torch.empty((8, h, w, cout), device=dev))
    d = (_lib.ConvDesc * 1)(); ops._fill_desc(d[0], x, pc, y, True, None, None, False, False)
    d[0].tune_wm, d[0].tune_sc, d[0].tune_wn = 6, 16, 2
    fl = 2.0 * 8 * h * w * cin * cout * 9
    for mask, what in [(0, "full"), (1, "no transform in loop"), (2, "no U glds in loop"), (4, "no halo load/store in loop"), (7, "MFMA + barriers only"), (8, "no MFMA")]:
        raw.cmk_debug_wino_ablation(mask)
        t = timeit(d[0])
        print("%-8s %-28s %7.3f ms  (%6.1f alg TF)" % (name, what, t, fl / t / 1e9))
    raw.cmk_debug_wino_ablation(0)
