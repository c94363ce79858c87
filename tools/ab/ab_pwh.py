"""Same-session timing of conv_pw builds on the fp16-split form (tune 12): ab_pwh.py [--rounds R] <lib.so> [<lib.so> ...]  (ablation builds: times only)"""
import sys, os, subprocess
SHAPES = [("OSA2_cat", 200, 320, 768, 256), ("OSA3_cat", 100, 160, 1056, 512), ("OSA4_cat", 50, 80, 1472, 768), ("OSA5_cat", 25, 40, 1888, 1024)]
if sys.argv[1] != "--one":
    args = sys.argv[1:]; rounds = 2
    if args[0] == "--rounds": rounds = int(args[1]); args = args[2:]
    res = {l: [] for l in args}
    for _ in range(rounds):
        for lib in args:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", lib], capture_output=True, text=True)
            try: res[lib].append([float(v) for v in r.stdout.strip().split("\n")[-1].split()])
            except Exception: print(lib, "FAILED", r.stdout[-300:], r.stderr[-600:]); res[lib].append([float("nan")] * len(SHAPES))
    for lib in args:
        best = [min(r[i] for r in res[lib]) for i in range(len(SHAPES))]
        print("%-16s" % os.path.basename(lib)[7:-3], " ".join("%s %.3f" % (n[0], b) for n, b in zip(SHAPES, best)), "| sum %.3f" % sum(best), flush=True)
    sys.exit(0)
import ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from centermask2_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[2])
from centermask2_amd import ops
ops.ALLOW_SPLIT_F16 = True
from centermask2_amd.ops import View
lib = _lib.load(); dev = torch.device("cuda:0"); out = []
for name, h, w, cin, cout in SHAPES:
    n = 8
    x = View(torch.randn((n, h, w, cin), device=dev).abs_()); pc = ops.PackedConv(torch.randn((cout, cin, 1, 1)) * 0.03, None, None, dev)
    y = View(torch.empty((n, h, w, cout), device=dev))
    d = (_lib.ConvDesc * 1)(); ops._fill_desc(d[0], x, pc, y, True, None, None, False, False)
    d[0].tune_wm, d[0].tune_sc, d[0].tune_wn = 12, 32, 4
    for _ in range(5): assert lib.cmk_conv2d_nhwc(ctypes.byref(d[0]), ops._stream()) == 0, lib.cmk_last_error()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): lib.cmk_conv2d_nhwc(ctypes.byref(d[0]), ops._stream())
    e1.record(); torch.cuda.synchronize()
    out.append(e0.elapsed_time(e1) / 10)
print(" ".join("%.4f" % v for v in out))
