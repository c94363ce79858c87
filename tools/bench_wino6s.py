"""F(4x4,3x3): 32 couts per workgroup, two workgroups per CU (tune 6/16, conv_wino6.hip) vs the shared-V form, 64 couts per workgroup
(tune 6/64, conv_wino6s.hip), per 3x3 layer shape of the model; same session, interleaved, best of `rounds`; checks that the two agree
bit for bit.  python tools/bench_wino6s.py [rounds] [v99]"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centermask2_amd import ops, _lib
from centermask2_amd.ops import View
dev = torch.device("cuda:0"); B = 8
SHAPES = [("stem_2", 400, 640, 64, 64), ("OSA2_x", 200, 320, 128, 128), ("OSA3_0", 100, 160, 256, 160), ("OSA3_x", 100, 160, 160, 160),
          ("OSA4_0", 50, 80, 512, 192), ("OSA4_x", 50, 80, 192, 192), ("OSA4_2_0", 50, 80, 768, 192), ("OSA5_0", 25, 40, 768, 224), ("OSA5_x", 25, 40, 224, 224),
          ("OSA5_2_0", 25, 40, 1024, 224), ("fpn_p3", 100, 160, 256, 256), ("fpn_p4", 50, 80, 256, 256), ("fpn_p5", 25, 40, 256, 256), ("roi", 14, 14, 256, 256),
          ("roi272", 14, 14, 272, 256)]
lib = _lib.load()
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
def timeit(d, it=8):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): lib.cmk_conv2d_nhwc(ctypes.byref(d), ops._stream())
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
print("%-9s %9s %8s %9s %8s %7s %6s" % ("layer", "w6 ms", "exeTF", "w6s ms", "exeTF", "w6/w6s", "equal"), flush=True)
tot = [0.0, 0.0]
for name, h, w, cin, cout in SHAPES:
    n = 400 if name.startswith("roi") else B
    roi = name.startswith("roi")
    x = View(torch.randn((n, h, w, cin), device=dev)); pc = ops.PackedConv(torch.randn((cout, cin, 3, 3)) * (2.0 / (9 * cin)) ** 0.5, None, None, dev)
    ys = [View(torch.empty((n, h, w, cout), device=dev)) for _ in range(2)]
    ds = []
    for k, tv in enumerate(((6, 16, 2 if roi else 1), (6, 64, 2 if roi else 1))):
        d = (_lib.ConvDesc * 1)(); ops._fill_desc(d[0], x, pc, ys[k], True, None, None, False, False)
        d[0].tune_wm, d[0].tune_sc, d[0].tune_wn = tv
        assert lib.cmk_conv2d_nhwc(ctypes.byref(d[0]), ops._stream()) == 0, lib.cmk_last_error()
        ds.append(d)
    torch.cuda.synchronize()
    eq = bool(torch.equal(ys[0].t, ys[1].t))
    best = [1e9, 1e9]
    for _ in range(rounds):
        for k in range(2): best[k] = min(best[k], timeit(ds[k][0]))
    ex = ops.executed_flops(9, 1, (6, 16, 2 if roi else 1), [(n, h, w)], pc.cin_pad, cout)
    tot[0] += best[0]; tot[1] += best[1]
    print("%-9s %9.3f %8.1f %9.3f %8.1f %7.2f %6s" % (name, best[0], ex / best[0] / 1e9, best[1], ex / best[1] / 1e9, best[0] / best[1], eq), flush=True)
print("sum w6 %.3f ms, w6s %.3f ms" % tuple(tot))
