"""Timeline of the direct fp16-split 3x3 kernel from an instrumented build (tools/ab/build_variant.sh sp3_trace conv_sp3.hip -DSP3_TRACE):
lane 0 of every wave of every 8th workgroup stamps the shader clock at the barrier and after every tap of the first six chunks.
trace_sp3.py <lib.so> [H W Cin Cout geo]"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from centermask2_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
from centermask2_amd import ops
ops.ALLOW_SPLIT_F16 = True
from centermask2_amd.ops import View
import numpy as np
lib = _lib.load(); dev = torch.device("cuda:0")
h, w, cin, cout, geo = [int(v) for v in sys.argv[2:7]] if len(sys.argv) > 6 else (200, 320, 128, 128, 0)
n = 8
x = View(torch.randn((n, h, w, cin), device=dev)); pc = ops.PackedConv(torch.randn((cout, cin, 3, 3)) * 0.05, None, None, dev)
y = View(torch.empty((n, h, w, cout), device=dev))
d = (_lib.ConvDesc * 1)(); ops._fill_desc(d[0], x, pc, y, True, None, None, False, False)
d[0].tune_wm, d[0].tune_sc, d[0].tune_wn = 11, 2, geo
th, tw, ct = ((8, 32, 128), (4, 32, 256), (16, 16, 128), (8, 16, 256))[geo]
tiles = n * -(-h // th) * -(-w // tw)
nwg = ((tiles + 7) // 8) * 8 * -(-cout // ct)
buf = torch.zeros(((nwg + 7) // 8 * 4 * 128,), dtype=torch.int64, device=dev)
for _ in range(3): assert lib.cmk_conv2d_nhwc(ctypes.byref(d[0]), ops._stream()) == 0, lib.cmk_last_error()
torch.cuda.synchronize()
d[0].splitk_ws = buf.data_ptr()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); assert lib.cmk_conv2d_nhwc(ctypes.byref(d[0]), ops._stream()) == 0; e1.record(); torch.cuda.synchronize()
print("launch %.3f ms, %d workgroups (%.2f rounds of 512 slots)" % (e0.elapsed_time(e1), nwg, nwg / 512.0))
t = buf.cpu().numpy().reshape(-1, 4, 128).astype(np.float64)
t = t[t[:, 0, 0] > 0]
rt = t[:, :, 122] - t[:, :, 123]
life = t[:, :, 121] - t[:, :, 0]
clk = np.median(life / (rt * 10.0))
print("sampled %d workgroups; shader clock %.2f GHz; life %.1f us (median; min %.1f max %.1f)" % (t.shape[0], clk, np.median(rt) / 100.0, rt.min() / 100.0, rt.max() / 100.0))
med = lambda a: np.median(a)
print("prologue %.0f cyc (%.2f us); loop %.0f cyc; epilogue %.0f cyc (%.2f us); ideal MFMA time of a chunk: 216 x 32 = 6912 cyc (x2 with the other wave of the SIMD)" % (
    med(t[:, :, 1] - t[:, :, 0]), med(t[:, :, 1] - t[:, :, 0]) / clk / 1e3, med(t[:, :, 120] - t[:, :, 1]), med(t[:, :, 121] - t[:, :, 120]), med(t[:, :, 121] - t[:, :, 120]) / clk / 1e3))
nch = min(6, cin // 16)
for c in range(nch):
    b = 2 + 11 * c
    taps = [med(t[:, :, b + 2 + k] - t[:, :, b + 1 + k]) for k in range(9)]
    print("chunk %d: barrier wait %5.0f | taps %s | chunk total %6.0f" % (c, med(t[:, :, b + 1] - t[:, :, b]), " ".join("%5.0f" % v for v in taps),
          med(t[:, :, b + 10] - t[:, :, b])))
# distribution of one mid tap and of the tap after the halo requests
b = 2 + 11 * min(2, nch - 1)
for k in (1, 3, 4, 5, 6, 7):
    v = t[:, :, b + 2 + k] - t[:, :, b + 1 + k]
    print("chunk 2 tap %d: p10 %5.0f p50 %5.0f p90 %5.0f max %6.0f" % (k, np.percentile(v, 10), np.percentile(v, 50), np.percentile(v, 90), v.max()))
