"""Micro-benchmark of conv_igemm on the V2-39 / FCOS layer shapes (HIP events on the launch stream)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centermask2_amd import ops
from centermask2_amd.ops import View

dev = torch.device("cuda:0")
B = 8
SHAPES = [  # name, H, W, Cin, Cout, k, stride
    ("stem_2", 400, 640, 64, 64, 3, 1), ("stem_3", 400, 640, 64, 128, 3, 2),
    ("OSA2_x", 200, 320, 128, 128, 3, 1), ("OSA2_cat", 200, 320, 768, 256, 1, 1),
    ("OSA3_0", 100, 160, 256, 160, 3, 1), ("OSA3_x", 100, 160, 160, 160, 3, 1), ("OSA3_cat", 100, 160, 1056, 512, 1, 1),
    ("OSA4_x", 50, 80, 192, 192, 3, 1), ("OSA4_cat", 50, 80, 1472, 768, 1, 1),
    ("OSA5_x", 25, 40, 224, 224, 3, 1), ("OSA5_cat", 25, 40, 2144, 1024, 1, 1),
    ("fcos_p3", 100, 160, 256, 256, 3, 1), ("fcos_p4", 50, 80, 256, 256, 3, 1), ("fcos_p5", 25, 40, 256, 256, 3, 1),
    ("cls_p3", 100, 160, 256, 80, 3, 1),
]
TUNE = len(sys.argv) > 1 and sys.argv[1] == "tune"
ops.set_autotune(TUNE)
print("%-10s %8s %8s %8s  %s" % ("layer", "ms", "TFLOP/s", "GB/s(alg)", "autotuned" if TUNE else "cost model"))
tot_ms = tot_fl = 0
for name, h, w, cin, cout, k, s in SHAPES:
    x = View(torch.randn((B, h, w, cin), device=dev))
    pc = ops.PackedConv(torch.randn((cout, cin, k, k)) * 0.05, None, None, dev, stride=s)
    y = ops.conv_out(x, pc, relu=True)
    for _ in range(2):
        ops.conv2d(x, pc, y, relu=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    it = 5
    e0.record()
    for _ in range(it):
        ops.conv2d(x, pc, y, relu=True)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / it
    ho, wo = y.t.shape[1], y.t.shape[2]
    fl = 2.0 * B * ho * wo * cin * cout * k * k
    by = 4.0 * (B * h * w * cin + B * ho * wo * cout + cin * cout * k * k)
    tv = [v for k_, v in ops.tuned_variants().items() if k_[2] == cin and k_[3] == cout and k_[7][0][1] == h]
    print("%-10s %8.3f %8.1f %8.0f  %s" % (name, ms, fl / ms / 1e9, by / ms / 1e6, tv[-1] if tv else ""))
    tot_ms += ms; tot_fl += fl
print("sum %.2f ms, %.1f TFLOP/s" % (tot_ms, tot_fl / tot_ms / 1e9))
