"""1x1 convs of the model: conv_igemm (tune_wm 1, the tuned WN) vs the pointwise GEMM kernel conv_pw (tune_wm 8, 256- and 128-pixel
workgroups), same session, interleaved; checks that they agree bit for bit.  python tools/bench_pw.py [rounds]"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centermask2_amd import ops, _lib
from centermask2_amd.ops import View
dev = torch.device("cuda:0"); B = 8
# (name, N, H, W, Cin, Cout, old variant)
SHAPES = [("OSA2_cat", B, 200, 320, 768, 256, (1, 32, 4)), ("OSA3_cat", B, 100, 160, 1056, 512, (1, 32, 4)),
          ("OSA4_cat0", B, 50, 80, 1472, 768, (1, 32, 4)), ("OSA4_cat1", B, 50, 80, 1728, 768, (1, 32, 4)),
          ("OSA5_cat0", B, 25, 40, 1888, 1024, (1, 32, 4)), ("OSA5_cat1", B, 25, 40, 2144, 1024, (1, 32, 4)),
          ("fpn_lat3", B, 100, 160, 512, 256, (1, 32, 2)), ("fpn_lat4", B, 50, 80, 768, 256, (1, 32, 2)),
          ("deconv", 400, 14, 14, 256, 1024, (1, 32, 2))]
lib = _lib.load()
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
def timeit(d, it=6):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): lib.cmk_conv2d_nhwc(ctypes.byref(d), ops._stream())
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
print("%-10s %9s %7s %9s %7s %9s %7s %6s" % ("layer", "igemm ms", "TF", "pw4 ms", "TF", "pw2 ms", "TF", "equal"), flush=True)
tot = [0.0, 0.0]
for name, n, h, w, cin, cout, old in SHAPES:
    x = View(torch.randn((n, h, w, cin), device=dev)); pc = ops.PackedConv(torch.randn((cout, cin, 1, 1)) * (2.0 / cin) ** 0.5, None, None, dev)
    ys = [View(torch.empty((n, h, w, cout), device=dev)) for _ in range(3)]
    ds = []
    for k, tv in enumerate((old, (8, 32, 4), (8, 32, 2))):
        d = (_lib.ConvDesc * 1)(); ops._fill_desc(d[0], x, pc, ys[k], True, None, None, False, False)
        d[0].tune_wm, d[0].tune_sc, d[0].tune_wn = tv
        assert lib.cmk_conv2d_nhwc(ctypes.byref(d[0]), ops._stream()) == 0, lib.cmk_last_error()
        ds.append(d)
    torch.cuda.synchronize()
    eq = torch.equal(ys[0].t, ys[1].t) and torch.equal(ys[0].t, ys[2].t)
    best = [1e9, 1e9, 1e9]
    for _ in range(rounds):
        for k in range(3): best[k] = min(best[k], timeit(ds[k][0]))
    fl = 2.0 * n * h * w * cin * cout
    tot[0] += best[0]; tot[1] += min(best[1], best[2])
    print("%-10s %9.3f %7.1f %9.3f %7.1f %9.3f %7.1f %6s" % (name, best[0], fl / best[0] / 1e9, best[1], fl / best[1] / 1e9, best[2], fl / best[2] / 1e9, eq), flush=True)
print("sum igemm %.3f ms, best pw %.3f ms" % tuple(tot))
