"""Run one 3x3 conv shape with a forced variant. usage: prof_wino.py H W Cin Cout wm sc wn [iters] [B]"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centermask2_amd import ops, _lib
from centermask2_amd.ops import View
h, w, cin, cout, wm, sc, wn = [int(v) for v in sys.argv[1:8]]
iters = int(sys.argv[8]) if len(sys.argv) > 8 else 3
B = int(sys.argv[9]) if len(sys.argv) > 9 else 8
dev = torch.device("cuda:0")
if wm in (10, 11): ops.ALLOW_SPLIT_BF16 = ops.ALLOW_SPLIT_F16 = True     # the opt-in split forms need their packing
lib = _lib.load()
x = View(torch.randn((B, h, w, cin), device=dev)); pc = ops.PackedConv(torch.randn((cout, cin, 3, 3)) * 0.05, None, None, dev)
y = View(torch.empty((B, h, w, cout), device=dev))
d = (_lib.ConvDesc * 1)(); ops._fill_desc(d[0], x, pc, y, True, None, None, False, False)
d[0].tune_wm, d[0].tune_sc, d[0].tune_wn = wm, sc, wn
for _ in range(iters + 1):
    assert lib.cmk_conv2d_nhwc(ctypes.byref(d[0]), ops._stream()) == 0, lib.cmk_last_error()
torch.cuda.synchronize()
