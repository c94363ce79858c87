"""Run one conv shape repeatedly (for rocprofv3 --pmc / --kernel-trace). usage: prof_conv.py H W Cin Cout k stride [iters] [B]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centermask2_amd import ops
from centermask2_amd.ops import View
h, w, cin, cout, k, s = [int(v) for v in sys.argv[1:7]]
iters = int(sys.argv[7]) if len(sys.argv) > 7 else 5
B = int(sys.argv[8]) if len(sys.argv) > 8 else 8
dev = torch.device("cuda:0")
x = View(torch.randn((B, h, w, cin), device=dev))
pc = ops.PackedConv(torch.randn((cout, cin, k, k)) * 0.05, None, None, dev, stride=s)
y = ops.conv_out(x, pc, relu=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    ops.conv2d(x, pc, y, relu=True)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
fl = 2.0 * B * y.t.shape[1] * y.t.shape[2] * cin * cout * k * k
print("%dx%d %d->%d k%d s%d: %.3f ms  %.1f TFLOP/s" % (h, w, cin, cout, k, s, ms, fl / ms / 1e9))
