"""3x3 convs: the best F(4x4,3x3) Winograd form (tune 6/16 | 6/64, fp32 MFMA) against the gather form of the pointwise kernel on bf16-split
products (tune 10/32/4, opt-in), same session, interleaved, best of `rounds`; max distance of each from a float64 conv on a sample.
python tools/bench_split3x3.py [rounds]"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centermask2_amd import ops, _lib
from centermask2_amd.ops import View
ops.ALLOW_SPLIT_BF16 = True
dev = torch.device("cuda:0"); B = 8
SHAPES = [("OSA2_x", 200, 320, 128, 128), ("fpn_p3", 100, 160, 256, 256), ("fpn_p4", 50, 80, 256, 256), ("fpn_p5", 25, 40, 256, 256), ("roi", 14, 14, 256, 256),
          ("c256_p3", 100, 160, 256, 128), ("c512", 50, 80, 512, 256)]
lib = _lib.load()
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
def timeit(d, it=8):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): lib.cmk_conv2d_nhwc(ctypes.byref(d), ops._stream())
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
print("%-9s %9s %9s %9s %8s %8s %10s %10s" % ("layer", "w6 ms", "w6s ms", "split ms", "dirTF", "best/spl", "err wino", "err split"), flush=True)
for name, h, w, cin, cout in SHAPES:
    roi = name.startswith("roi"); n = 400 if roi else B
    x = View(torch.randn((n, h, w, cin), device=dev)); wt = torch.randn((cout, cin, 3, 3)) * (2.0 / (9 * cin)) ** 0.5
    pc = ops.PackedConv(wt, None, None, dev)
    tvs = ((6, 16, 2 if roi else 1), (6, 64, 2 if roi else 1), (10, 32, 4))
    ys = [View(torch.empty((n, h, w, cout), device=dev)) for _ in tvs]
    ds = []
    for k, tv in enumerate(tvs):
        d = (_lib.ConvDesc * 1)(); ops._fill_desc(d[0], x, pc, ys[k], True, None, None, False, False)
        d[0].tune_wm, d[0].tune_sc, d[0].tune_wn = tv
        assert lib.cmk_conv2d_nhwc(ctypes.byref(d[0]), ops._stream()) == 0, (tv, lib.cmk_last_error())
        ds.append(d)
    torch.cuda.synchronize()
    ref = torch.nn.functional.conv2d(x.t[:1].permute(0, 3, 1, 2).double(), wt.to(dev).double(), padding=1).relu().permute(0, 2, 3, 1)
    errs = [float((ys[k].t[:1].double() - ref).abs().max()) for k in range(3)]
    best = [1e9] * 3
    for _ in range(rounds):
        for k in range(3): best[k] = min(best[k], timeit(ds[k][0]))
    fl = 2.0 * n * h * w * 9 * cin * cout
    print("%-9s %9.3f %9.3f %9.3f %8.1f %8.2f %10.2e %10.2e" % (name, best[0], best[1], best[2], fl / best[2] / 1e9, min(best[0], best[1]) / best[2], max(errs[0], errs[1]), errs[2]), flush=True)
