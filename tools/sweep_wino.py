import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
WM = 5          # the Winograd kernel
from centermask2_amd import ops, _lib
from centermask2_amd.ops import View
lib = _lib.load()
g = torch.Generator().manual_seed(1)
for (n, h, w, cin, cout) in [(1,16,16,64,64),(1,16,16,128,64),(1,16,16,256,64),(1,16,16,256,256),(1,16,16,64,256),(2,37,45,256,128),(1,8,16,256,64),(1,8,16,96,64),(1,8,16,80,64),(1,8,16,48,64)]:
    x = torch.randn((n, cin, h, w), generator=g); wt = torch.randn((cout, cin, 3, 3), generator=g) * (2.0/(cin*9))**0.5
    ref = F.conv2d(x, wt, None, padding=1)
    pc = ops.PackedConv(wt, None, None, "cuda"); xv = ops.as_view(x.cuda())
    errs = []
    for rep in range(3):
        y = View(torch.full((n, h, w, cout), -5.0, device="cuda"))
        d = (_lib.ConvDesc * 1)(); ops._fill_desc(d[0], xv, pc, y, False, None, None, False, False)
        d[0].tune_wm, d[0].tune_sc, d[0].tune_wn = WM, 16, 2
        assert lib.cmk_conv2d_nhwc(ctypes.byref(d[0]), ops._stream()) == 0
        torch.cuda.synchronize()
        e = (y.nchw().cpu() - ref).abs()
        errs.append(float(e.max()))
    bad = (e > 1e-2)
    print((n,h,w,cin,cout), "max err per rep", ["%.2e" % v for v in errs], "bad frac %.4f" % float(bad.float().mean()),
          "bad channels" , sorted(set(torch.nonzero(bad)[:,1].tolist()))[:8] if bad.any() else "")
