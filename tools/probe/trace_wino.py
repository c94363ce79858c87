"""Poor man's thread trace of the 2-WG Winograd kernel (needs a library built with -DCMK_WINO_TRACE).
usage: trace_wino.py <libcmk_TRACE.so> H W Cin Cout [N]"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from centermask2_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
from centermask2_amd import ops
from centermask2_amd.ops import View
h, w, cin, cout = [int(v) for v in sys.argv[2:6]]
n = int(sys.argv[6]) if len(sys.argv) > 6 else 8
lib = _lib.load()
dev = torch.device("cuda:0")
x = View(torch.randn((n, h, w, cin), device=dev)); pc = ops.PackedConv(torch.randn((cout, cin, 3, 3)) * 0.05, None, None, dev)
y = View(torch.empty((n, h, w, cout), device=dev))
d = (_lib.ConvDesc * 1)(); ops._fill_desc(d[0], x, pc, y, True, None, None, False, False)
d[0].tune_wm, d[0].tune_sc, d[0].tune_wn = 6, 16, 2
tiles = n * ((h + 7) // 8) * ((w + 15) // 16)
blocks = (tiles + 7) // 8 * 8 * ((cout + 63) // 64)
buf = torch.zeros((blocks, 4, 160), dtype=torch.int64, device=dev)
d[0].gn_ws = buf.data_ptr()
for _ in range(10):
    assert lib.cmk_conv2d_nhwc(ctypes.byref(d[0]), ops._stream()) == 0
torch.cuda.synchronize()
buf.zero_()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); lib.cmk_conv2d_nhwc(ctypes.byref(d[0]), ops._stream()); e1.record(); torch.cuda.synchronize()
print("launch %.3f ms (instrumented), %d blocks" % (e0.elapsed_time(e1), blocks))
t = buf.cpu()
cnt = t[:, :, 159]
valid = cnt[:, 0] > 0
t = t[valid].double()
nb = t.shape[0]
nsteps = (cin + 15) // 16 * 4
TICK = 10.0  # ns per s_memrealtime tick (100 MHz)
start, loop_in = t[:, :, 1], t[:, :, 2]
arrive = t[:, :, 3:3 + 2 * nsteps:2]; leave = t[:, :, 4:4 + 2 * nsteps:2]
loop_out = t[:, :, 3 + 2 * nsteps]; end = t[:, :, 8 + 2 * nsteps]
e = [t[:, :, 3 + 2 * nsteps + i] for i in range(6)]
print("epilogue pieces (ns): barrier1 %.0f | output transform + LDS exchange writes %.0f | barrier2 %.0f | exchange reads + stores issued %.0f | stores retired %.0f" % tuple(
      ((e[i + 1] - e[i]) * TICK).mean() for i in range(5)))
wait = (leave - arrive) * TICK
work = torch.cat([(arrive[:, :, 1:] - leave[:, :, :-1]), (loop_out - leave[:, :, -1])[:, :, None]], 2) * TICK
print("blocks traced %d, steps %d" % (nb, nsteps))
print("per wave (ns): total %.0f | prologue %.0f | loop %.0f | epilogue %.0f" % (((end - start) * TICK).mean(), ((loop_in - start) * TICK).mean(),
      ((loop_out - loop_in) * TICK).mean(), ((end - loop_out) * TICK).mean()))
print("per step (ns): barrier wait mean %.0f (p50 %.0f p90 %.0f) | work between barriers mean %.0f (p50 %.0f p90 %.0f)" % (
      wait.mean(), wait.flatten().quantile(0.5), wait.flatten().quantile(0.9), work.mean(), work.flatten().quantile(0.5), work.flatten().quantile(0.9)))
for g in range(4):
    print("  step g=%d: wait %.0f  work %.0f" % (g, wait[:, :, g::4].mean(), work[:, :, g::4].mean()))
# launch-level: when do blocks start/finish
t0 = start.min()
print("first start -> last end: %.1f us; mean block duration %.1f us" % ((end.max() - t0) * TICK / 1e3, ((end.max(1).values - start.min(1).values) * TICK).mean() / 1e3))
# busy fraction estimate per wave inside loop: MFMA time if alone = 16 * 64 cycles per step
print("ideal MFMA issue per step at 2.4 GHz, pipe alone: %.0f ns; shared by two waves: %.0f ns" % (1024 / 2.4, 2048 / 2.4))
