"""Where a wave of the pointwise GEMM kernel spends its time, from an instrumented build (tools/ab/build_variant.sh pw_trace
centermask2_amd/csrc/conv_pw.hip -DPW_TRACE): lane 0 of every wave of every 16th workgroup stamps the shader clock in front of and
behind every chunk's barrier.   trace_pw.py <lib.so> [H W Cin Cout wn]"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from centermask2_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
from centermask2_amd import ops
from centermask2_amd.ops import View
import numpy as np
lib = _lib.load(); dev = torch.device("cuda:0")
h, w, cin, cout, wn = [int(v) for v in sys.argv[2:7]] if len(sys.argv) > 6 else (200, 320, 768, 256, 4)
n = 8
x = View(torch.randn((n, h, w, cin), device=dev)); pc = ops.PackedConv(torch.randn((cout, cin, 1, 1)) * 0.05, None, None, dev)
y = View(torch.empty((n, h, w, cout), device=dev))
d = (_lib.ConvDesc * 1)(); ops._fill_desc(d[0], x, pc, y, True, None, None, False, False)
d[0].tune_wm, d[0].tune_sc, d[0].tune_wn = 8, 32, wn
tiles = -(-(n * h * w) // (64 * wn))
nwg = ((tiles + 7) // 8) * 8 * (cout // 128)
buf = torch.zeros(((nwg + 15) // 16 * 4 * 64,), dtype=torch.int64, device=dev)
for _ in range(3): assert lib.cmk_conv2d_nhwc(ctypes.byref(d[0]), ops._stream()) == 0, lib.cmk_last_error()
torch.cuda.synchronize()
d[0].splitk_ws = buf.data_ptr()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); assert lib.cmk_conv2d_nhwc(ctypes.byref(d[0]), ops._stream()) == 0; e1.record(); torch.cuda.synchronize()
print("launch %.3f ms, %d workgroups (%.2f rounds of 512)" % (e0.elapsed_time(e1), nwg, nwg / 512.0))
t = buf.cpu().numpy().reshape(-1, 4, 64).astype(np.float64)
t = t[t[:, 0, 0] > 0]
rt = t[:, :, 62] - t[:, :, 63]
life = t[:, :, 59] - t[:, :, 0]
clk = life / (rt * 10.0)
ck = np.median(clk)
print("sampled %d workgroups; shader clock %.2f GHz; life %.1f us median (min %.1f max %.1f)" % (t.shape[0], ck, np.median(rt) / 100, rt.min() / 100, rt.max() / 100))
nst = min(cin // 16, 28)
reach = t[:, :, 2:2 + 2 * nst:2]; passed = t[:, :, 3:3 + 2 * nst:2]
wait = passed - reach                                  # cycles in the barrier
period = reach[:, :, 1:] - reach[:, :, :-1]
mt_mfma = 16 * wn * 64                                 # MFMA pipe cycles of one wave per chunk
print("prologue %.0f cycles (%.2f us); first barrier wait %.0f" % (np.median(t[:, :, 1] - t[:, :, 0]), np.median(t[:, :, 1] - t[:, :, 0]) / ck / 1e3, np.median(wait[:, :, 0])))
print("chunk period: median %.0f cycles, p10 %.0f, p90 %.0f; MFMA cycles of the two waves of a SIMD per chunk = %d" % (
    np.median(period), np.percentile(period, 10), np.percentile(period, 90), 2 * mt_mfma))
print("barrier wait per chunk: median %.0f cycles, mean %.0f, p90 %.0f  (%.1f %% of the period)" % (
    np.median(wait[:, :, 1:]), wait[:, :, 1:].mean(), np.percentile(wait[:, :, 1:], 90), 100 * wait[:, :, 1:].mean() / period.mean()))
print("per-chunk median period:", " ".join("%.0f" % np.median(period[:, :, i]) for i in range(period.shape[2])))
print("per-chunk median wait:  ", " ".join("%.0f" % np.median(wait[:, :, i]) for i in range(wait.shape[2])))
# skew between the waves of a workgroup when they reach a barrier
skew = reach.max(axis=1) - reach.min(axis=1)
print("skew between the first and the last wave of a workgroup at a barrier: median %.0f cycles, p90 %.0f" % (np.median(skew[:, 1:]), np.percentile(skew[:, 1:], 90)))
print("epilogue %.0f cycles (%.2f us); loop %.1f us; life %.1f us" % (np.median(t[:, :, 59] - t[:, :, 58]), np.median(t[:, :, 59] - t[:, :, 58]) / ck / 1e3,
      np.median(t[:, :, 58] - t[:, :, 1]) / ck / 1e3, np.median(life) / ck / 1e3))
print("launch time / rounds = %.1f us per slot-life" % (e0.elapsed_time(e1) * 1e3 / (nwg / 512.0)))
