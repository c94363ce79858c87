"""Same-session timing of conv_sp3 builds (ablations, variants): ab_sp3.py [--rounds R] <lib.so> [<lib.so> ...]
One subprocess per (round, library), interleaved, best-of per shape.  Results of ablation builds are wrong by design; only times count."""
import sys, os, subprocess
SHAPES = [("OSA2_x", 200, 320, 128, 128, 0), ("OSA3_x", 100, 160, 160, 160, 1), ("fpn_p3", 100, 160, 256, 256, 1), ("fpn_p4", 50, 80, 256, 256, 3), ("fpn_p5", 25, 40, 256, 256, 3), ("p6", 13, 20, 256, 256, 3), ("roi", 14, 14, 256, 256, 3)]
if sys.argv[1] != "--one":
    args = sys.argv[1:]
    rounds, sc = 2, "64"
    while args and args[0].startswith("--"):
        if args[0] == "--rounds": rounds = int(args[1])
        if args[0] == "--sc": sc = args[1]
        args = args[2:]
    res = {l: [] for l in args}
    for _ in range(rounds):
        for lib in args:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", lib, sc], capture_output=True, text=True)
            try:
                res[lib].append([float(v) for v in r.stdout.strip().split("\n")[-1].split()])
            except Exception:
                print(lib, "FAILED", r.stdout[-300:], r.stderr[-600:]); res[lib].append([float("nan")] * len(SHAPES))
    for lib in args:
        best = [min(r[i] for r in res[lib]) for i in range(len(SHAPES))]
        print("%-22s" % os.path.basename(lib)[7:-3], " ".join("%s %.3f" % (n[0], b) for n, b in zip(SHAPES, best)), "| sum %.3f" % sum(best), flush=True)
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
sc = int(sys.argv[3])
for name, h, w, cin, cout, geo in SHAPES:
    n = 400 if name == "roi" else 8
    x = View(torch.randn((n, h, w, cin), device=dev)); pc = ops.PackedConv(torch.randn((cout, cin, 3, 3)) * 0.05, None, None, dev)
    y = View(torch.empty((n, h, w, cout), device=dev))
    d = (_lib.ConvDesc * 1)(); ops._fill_desc(d[0], x, pc, y, True, None, None, False, False)
    d[0].tune_wm, d[0].tune_sc, d[0].tune_wn = 11, 2, geo
    for _ in range(10): assert lib.cmk_conv2d_nhwc(ctypes.byref(d[0]), ops._stream()) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): lib.cmk_conv2d_nhwc(ctypes.byref(d[0]), ops._stream())
    e1.record(); torch.cuda.synchronize()
    out.append(e0.elapsed_time(e1) / 20)
print(" ".join("%.4f" % v for v in out))
