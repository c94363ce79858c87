"""Are the two workgroups of a CU in lock-step?  Instrumented build with every workgroup traced (-DW6_TRACE -DW6_TRACE_EVERY=1):
start/end real-time stamps and HW_ID/XCC_ID per workgroup -> per-CU timelines.   phase_wino6.py <lib.so> [H W Cin Cout]"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
from centermask2_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
from centermask2_amd import ops
from centermask2_amd.ops import View
lib = _lib.load(); dev = torch.device("cuda:0")
h, w, cin, cout = [int(v) for v in sys.argv[2:6]] if len(sys.argv) > 5 else (200, 320, 128, 128)
n = 8
x = View(torch.randn((n, h, w, cin), device=dev)); pc = ops.PackedConv(torch.randn((cout, cin, 3, 3)) * 0.05, None, None, dev)
y = View(torch.empty((n, h, w, cout), device=dev))
d = (_lib.ConvDesc * 1)(); ops._fill_desc(d[0], x, pc, y, True, None, None, False, False)
d[0].tune_wm, d[0].tune_sc, d[0].tune_wn = 6, 16, 1
tiles = n * -(-h // 12) * -(-w // 40)
nwg = ((tiles + 7) // 8) * 8 * -(-cout // 32)
buf = torch.zeros((nwg * 4 * 64,), dtype=torch.int64, device=dev)
for _ in range(3): assert lib.cmk_conv2d_nhwc(ctypes.byref(d[0]), ops._stream()) == 0
torch.cuda.synchronize()
d[0].splitk_ws = buf.data_ptr()
assert lib.cmk_conv2d_nhwc(ctypes.byref(d[0]), ops._stream()) == 0
torch.cuda.synchronize()
t = buf.cpu().numpy().reshape(-1, 4, 64)
t = t[t[:, 0, 63] > 0]
start, end = t[:, 0, 63].astype(np.float64) / 100.0, t[:, 0, 62].astype(np.float64) / 100.0     # us
hw, xcc = t[:, 0, 61], t[:, 0, 60] & 15
wave_id, simd, cu, sh, se = hw & 15, (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
t0 = start.min()
print("workgroups traced", len(start), "distinct CUs", len(np.unique(key)), "wave_id values of wave 0:", np.unique(wave_id, return_counts=True))
# per CU: overlap structure
offs = []
for k in np.unique(key)[:6]:
    idx = np.where(key == k)[0]
    o = idx[np.argsort(start[idx])]
    print("CU key %d: %d workgroups; first starts (us since launch, wave slot):" % (k, len(o)), " ".join("%.1f/%d" % (start[i] - t0, wave_id[i]) for i in o[:8]))
for k in np.unique(key):
    idx = np.where(key == k)[0]
    for slot in np.unique(wave_id[idx]):
        pass
    o = idx[np.argsort(start[idx])]
    # phase offset between consecutive starts on the same CU relative to the median life
    life = np.median(end[o] - start[o])
    ds = np.diff(start[o])
    offs += list((ds % life) / life)
offs = np.array(offs)
print("median life %.1f us; start-to-start gaps on a CU modulo the life, as a fraction (0 or 1 = lock-step, 0.5 = alternating):" % np.median(end - start))
print(" histogram over 10 bins:", np.histogram(offs, bins=10, range=(0, 1))[0].tolist())
