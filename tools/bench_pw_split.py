"""The pointwise GEMM kernel's forms on the model's 1x1 shapes: fp32 MFMA (tune 8), three bf16 pieces / six products (10), two fp16 pieces / three
products (12); same session, interleaved, best of `rounds`; max distance from a float64 conv of image 0.  python tools/bench_pw_split.py [rounds]"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centermask2_amd import ops, _lib
from centermask2_amd.ops import View
ops.ALLOW_SPLIT_BF16 = ops.ALLOW_SPLIT_F16 = True
dev = torch.device("cuda:0"); B = 8
SHAPES = [("OSA2_cat", 200, 320, 768, 256), ("OSA3_cat", 100, 160, 1056, 512), ("OSA4_cat", 50, 80, 1472, 768), ("OSA4_2cat", 50, 80, 1728, 768),
          ("OSA5_cat", 25, 40, 1888, 1024), ("lat3", 100, 160, 512, 256), ("deconv", 14, 14, 256, 1024)]
lib = _lib.load()
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
def timeit(d, it=8):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): lib.cmk_conv2d_nhwc(ctypes.byref(d), ops._stream())
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
print("%-10s %8s %8s %8s | %7s %7s %7s | %9s %9s %9s" % ("layer", "f32 ms", "bf16x6", "fp16x3", "TF f32", "TF b6", "TF h3", "err f32", "err b6", "err h3"), flush=True)
for name, h, w, cin, cout in SHAPES:
    n = 400 if name == "deconv" else B
    x = View(torch.randn((n, h, w, cin), device=dev).abs_()); wt = torch.randn((cout, cin, 1, 1)) * (2.0 / cin) ** 0.5
    pc = ops.PackedConv(wt, torch.rand(cout) + 0.5, torch.randn(cout) * 0.1, dev)
    ref = torch.nn.functional.conv2d(x.t[:1].permute(0, 3, 1, 2).double(), wt.to(dev).double())
    ref = (ref * pc.scale.double()[None, :, None, None] + pc.shift.double()[None, :, None, None]).relu().permute(0, 2, 3, 1)
    best, errs = [], []
    for tv in ((8, 32, 4), (10, 32, 4), (12, 32, 4)):
        y = View(torch.full((n, h, w, cout), float("nan"), device=dev))
        d = (_lib.ConvDesc * 1)(); ops._fill_desc(d[0], x, pc, y, True, None, None, False, False)
        d[0].tune_wm, d[0].tune_sc, d[0].tune_wn = tv
        assert lib.cmk_conv2d_nhwc(ctypes.byref(d[0]), ops._stream()) == 0, (tv, lib.cmk_last_error())
        torch.cuda.synchronize()
        errs.append(float((y.t[:1].double() - ref).abs().max()))
        best.append(min(timeit(d[0]) for _ in range(rounds)))
    fl = 2.0 * n * h * w * cin * cout
    print("%-10s %8.3f %8.3f %8.3f | %7.1f %7.1f %7.1f | %9.2e %9.2e %9.2e" % ((name,) + tuple(best) + tuple(fl / b / 1e9 for b in best) + tuple(errs)), flush=True)
