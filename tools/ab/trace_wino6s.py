"""Timeline of the shared-V F(4x4,3x3) kernel from an instrumented build (tools/ab/build_variant.sh strace conv_wino6s.hip -DW6S_TRACE):
lane 0 of every wave of every 8th workgroup stamps the shader clock around T / barrier / M of the first 9 periods and around the epilogue.
trace_wino6s.py <lib.so> [H W Cin Cout]"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from centermask2_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
from centermask2_amd import ops
from centermask2_amd.ops import View
import numpy as np
lib = _lib.load(); dev = torch.device("cuda:0")
h, w, cin, cout = [int(v) for v in sys.argv[2:6]] if len(sys.argv) > 5 else (200, 320, 128, 128)
n = 8
x = View(torch.randn((n, h, w, cin), device=dev)); pc = ops.PackedConv(torch.randn((cout, cin, 3, 3)) * 0.05, None, None, dev)
y = View(torch.empty((n, h, w, cout), device=dev))
d = (_lib.ConvDesc * 1)(); ops._fill_desc(d[0], x, pc, y, True, None, None, False, False)
d[0].tune_wm, d[0].tune_sc, d[0].tune_wn = 6, 64, 1
tiles = n * -(-h // 12) * -(-w // 40)
nwg = ((tiles + 7) // 8) * 8 * -(-(-(-cout // 32)) // 2)
buf = torch.zeros(((nwg + 7) // 8 * 8 * 64,), dtype=torch.int64, device=dev)
for _ in range(3): assert lib.cmk_conv2d_nhwc(ctypes.byref(d[0]), ops._stream()) == 0
torch.cuda.synchronize()
d[0].splitk_ws = buf.data_ptr()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); assert lib.cmk_conv2d_nhwc(ctypes.byref(d[0]), ops._stream()) == 0; e1.record(); torch.cuda.synchronize()
print("launch %.3f ms, %d workgroups (%.2f rounds of 256)" % (e0.elapsed_time(e1), nwg, nwg / 256.0))
t = buf.cpu().numpy().reshape(-1, 8, 64).astype(np.float64)
t = t[t[:, 0, 0] > 0]
rt = (t[:, :, 62] - t[:, :, 63])                   # 100 MHz real-time ticks over the workgroup's life
life = t[:, :, 49] - t[:, :, 0]
clk = np.median(life / (rt * 10.0))                # shader cycles per ns
print("sampled %d workgroups; shader clock %.2f GHz (median); life %.1f us (median, min %.1f max %.1f)" % (t.shape[0], clk, np.median(rt) / 100.0, rt.min() / 100.0, rt.max() / 100.0))
def cyc(a): return np.median(a)
for name, ws in (("waves 0-3 (T first)", slice(0, 4)), ("waves 4-7 (M first)", slice(4, 8))):
    tt = t[:, ws, :]
    print(name)
    print("  prologue %.0f cyc = %.2f us; first barrier wait %.0f" % (cyc(tt[:, :, 1] - tt[:, :, 0]), cyc(tt[:, :, 1] - tt[:, :, 0]) / clk / 1e3, cyc(tt[:, :, 2] - tt[:, :, 1])))
    for c in range(2, 9):
        b = 3 + 4 * c
        prev_end = tt[:, :, b - 1]
        print("  period %d: gap/barrier before T %5.0f | T %5.0f | T->M (barrier for ct=1) %5.0f | M %5.0f | total since previous M end %5.0f" % (
            c, cyc(tt[:, :, b] - prev_end), cyc(tt[:, :, b + 1] - tt[:, :, b]), cyc(tt[:, :, b + 2] - tt[:, :, b + 1]), cyc(tt[:, :, b + 3] - tt[:, :, b + 2]),
            cyc(tt[:, :, b + 3] - prev_end)))
    print("  loop end -> epilogue barrier passed %.0f; epilogue %.0f cyc = %.2f us; life %.0f cyc = %.2f us" % (
        cyc(tt[:, :, 43] - tt[:, :, 42]), cyc(tt[:, :, 49] - tt[:, :, 43]), cyc(tt[:, :, 49] - tt[:, :, 43]) / clk / 1e3, cyc(tt[:, :, 49] - tt[:, :, 0]), cyc(tt[:, :, 49] - tt[:, :, 0]) / clk / 1e3))
print("launch time / rounds = %.1f us per workgroup slot" % (e0.elapsed_time(e1) * 1e3 / (nwg / 256.0)))
