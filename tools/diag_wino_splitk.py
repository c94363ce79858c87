"""Max abs error against a float64 convolution: F(4x4) unsplit, split-K 2/4, F(2x2), direct — one shape.  python tools/diag_wino_splitk.py"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from centermask2_amd import ops, _lib
from centermask2_amd.ops import View
dev = torch.device("cuda:0"); lib = _lib.load()
for (n, h, w, cin, cout) in [(2, 25, 40, 1024, 224), (2, 50, 80, 512, 192)]:
    g = torch.Generator().manual_seed(5)
    x = torch.randn((n, cin, h, w), generator=g); wt = torch.randn((cout, cin, 3, 3), generator=g) * (2.0 / (9 * cin)) ** 0.5
    ref = F.conv2d(x.double(), wt.double(), None, padding=1)
    ref32 = F.conv2d(x, wt, None, padding=1)
    pc = ops.PackedConv(wt, None, None, dev); xv = ops.as_view(x.to(dev))
    print("shape", (n, h, w, cin, cout), "max|ref| %.2f; torch fp32 CPU vs fp64: %.2e" % (float(ref.abs().max()), float((ref32.double() - ref).abs().max())))
    for tv in [(1, 16, 1), (5, 16, 2), (6, 16, 1), (6, 16, 1, 2), (6, 16, 1, 4), (6, 64, 1)]:
        y = View(torch.empty((n, h, w, cout), device=dev))
        d = (_lib.ConvDesc * 1)(); ops._fill_desc(d[0], xv, pc, y, False, None, None, False, False)
        ws = ops._set_variant(d, 1, tv)
        rc = lib.cmk_conv2d_nhwc(ctypes.byref(d[0]), ops._stream()); torch.cuda.synchronize()
        if rc: print("  ", tv, "refused"); continue
        e = (y.nchw().cpu().double() - ref).abs()
        print("  ", tv, "max abs err %.2e  rms %.2e" % (float(e.max()), float(e.pow(2).mean().sqrt())))
