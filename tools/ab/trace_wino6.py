"""Timeline of the F(4x4,3x3) kernel from an instrumented build (tools/ab/build_variant.sh trace <src> -DW6_TRACE):
lane 0 of every wave of every 16th workgroup stamps the shader clock at the prologue end, at every period's barrier and around the
epilogue rounds.   trace_wino6.py <lib.so> [H W Cin Cout]"""
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
d[0].tune_wm, d[0].tune_sc, d[0].tune_wn = 6, 16, 1
tiles = n * -(-h // 12) * -(-w // 40)
nwg = ((tiles + 7) // 8) * 8 * -(-cout // 32)
buf = torch.zeros(((nwg + 15) // 16 * 4 * 64,), dtype=torch.int64, device=dev)
for _ in range(3): assert lib.cmk_conv2d_nhwc(ctypes.byref(d[0]), ops._stream()) == 0
torch.cuda.synchronize()
d[0].splitk_ws = buf.data_ptr()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); assert lib.cmk_conv2d_nhwc(ctypes.byref(d[0]), ops._stream()) == 0; e1.record(); torch.cuda.synchronize()
print("launch %.3f ms, %d workgroups" % (e0.elapsed_time(e1), nwg))
t = buf.cpu().numpy().reshape(-1, 4, 64).astype(np.float64)
ok = t[:, 0, 0] > 0
t = t[ok]
nch = cin // 8
rt = (t[:, :, 62] - t[:, :, 63])                   # 100 MHz real-time ticks over the workgroup's life
life = t[:, :, 49] - t[:, :, 0]
clk = life / (rt * 10.0)                           # shader cycles per ns
print("sampled %d workgroups; shader clock %.2f GHz (median), life %.1f us (median), min/max %.1f/%.1f" % (
    t.shape[0], np.median(clk), np.median(rt) / 100.0, rt.min() / 100.0, rt.max() / 100.0))
def us(a): return np.median(a) / np.median(clk) / 1e3
print("prologue            %.2f us" % us(t[:, :, 1] - t[:, :, 0]))
print("first barrier wait  %.2f us" % us(t[:, :, 2] - t[:, :, 1]))
per = t[:, :, 3:2 + nch] - t[:, :, 2:1 + nch]
print("period (median over periods and waves) %.2f us = %.0f cycles; p10 %.0f p90 %.0f cycles; ideal at 2 WG/CU = 4608 cycles" % (
    us(per), np.median(per), np.percentile(per, 10), np.percentile(per, 90)))
print("per-period medians (cycles):", " ".join("%.0f" % np.median(per[:, :, i]) for i in range(per.shape[2])))
print("last period + loop exit   %.2f us" % us(t[:, :, 42] - t[:, :, 1 + nch]))
print("loop-end barrier    %.2f us" % us(t[:, :, 43] - t[:, :, 42]))
print("round 0 write %.2f | barrier %.2f | finish+store %.2f us" % (us(t[:, :, 44] - t[:, :, 43]), us(t[:, :, 45] - t[:, :, 44]), us(t[:, :, 46] - t[:, :, 45])))
print("round 1 write %.2f | barrier %.2f | finish+store %.2f us" % (us(t[:, :, 47] - t[:, :, 46]), us(t[:, :, 48] - t[:, :, 47]), us(t[:, :, 49] - t[:, :, 48])))
print("epilogue total      %.2f us; loop total %.2f us; life %.2f us" % (us(t[:, :, 49] - t[:, :, 42]), us(t[:, :, 42] - t[:, :, 2]), us(life)))
if t[:, :, 50].min() > 0:
    names = ["barrier -> half-step 0 issued (reads, transform, 6 MFMAs)", "half-step 1 issued", "pass 1 (wait halo, transform, LDS writes)", "halo loads issued",
             "step 1 issued (12 MFMAs)", "step 2 issued (12 MFMAs)"]
    print("inside period 8 (cycles, median over waves): " + "; ".join("%s %.0f" % (nm, np.median(t[:, :, 51 + i] - t[:, :, 50 + i])) for i, nm in enumerate(names)))
    print("   period-8 total stamped %.0f cycles" % np.median(t[:, :, 56] - t[:, :, 50]))
# how the workgroups of one CU overlap cannot be read from here; the launch time / rounds gives the per-slot life:
print("launch time / (workgroups / 512 slots) = %.1f us per slot-life" % (e0.elapsed_time(e1) * 1e3 / (nwg / 512.0)))
