"""Stride-2 3x3 convs of the model: conv_igemm (the tuned variants) vs the gather form on the pointwise GEMM kernel (tune_wm 9).
python tools/bench_ga.py [rounds]"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centermask2_amd import ops, _lib
from centermask2_amd.ops import View
dev = torch.device("cuda:0")
SHAPES = [("stem_3", 8, 400, 640, 64, 128, 2, (1, 16, 4)), ("miou_c4", 400, 14, 14, 256, 256, 2, (7, 32, 4, 4)), ("p6", 8, 25, 40, 256, 256, 2, (7, 32, 2, 8))]
lib = _lib.load()
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
def timeit(d, it=6):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): lib.cmk_conv2d_nhwc(ctypes.byref(d), ops._stream())
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
print("%-8s %9s %7s %9s %7s %9s %7s %6s" % ("layer", "igemm ms", "TF", "ga4 ms", "TF", "ga2 ms", "TF", "equal"), flush=True)
for name, n, h, w, cin, cout, stride, old in SHAPES:
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    x = View(torch.randn((n, h, w, cin), device=dev)); pc = ops.PackedConv(torch.randn((cout, cin, 3, 3)) * (2.0 / (9 * cin)) ** 0.5, None, None, dev, stride=stride)
    ys = [View(torch.empty((n, ho, wo, cout), device=dev)) for _ in range(3)]
    ds, keep = [], []
    for k, tv in enumerate((old, (9, 32, 4), (9, 32, 2))):
        d = (_lib.ConvDesc * 1)(); ops._fill_desc(d[0], x, pc, ys[k], True, None, None, False, False)
        keep.append(ops._set_variant(d, 1, tv))
        rc = lib.cmk_conv2d_nhwc(ctypes.byref(d[0]), ops._stream())
        ds.append(d if rc == 0 else None)
    torch.cuda.synchronize()
    eq = all(d is None or torch.equal(ys[0].t, ys[k].t) for k, d in enumerate(ds))
    best = [1e9, 1e9, 1e9]
    for _ in range(rounds):
        for k in range(3):
            if ds[k] is not None: best[k] = min(best[k], timeit(ds[k][0]))
    fl = 2.0 * n * ho * wo * cin * cout * 9
    print("%-8s %9.3f %7.1f %9.3f %7.1f %9.3f %7.1f %6s" % (name, best[0], fl / best[0] / 1e9, best[1], fl / best[1] / 1e9, best[2], fl / best[2] / 1e9, eq), flush=True)
