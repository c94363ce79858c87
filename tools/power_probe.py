"""Is a conv kernel clock/power-limited?  Runs one conv shape back to back for a few seconds while sampling rocm-smi (power, sclk).
usage: power_probe.py H W Cin Cout wm[:wn] [seconds]     (wm 1 or 8 with :wn = a 1x1 conv with that variant forced)"""
import sys, os, ctypes, subprocess, threading, time, re
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centermask2_amd import ops, _lib
from centermask2_amd.ops import View
h, w, cin, cout = [int(v) for v in sys.argv[1:5]]
wm, _, wn = sys.argv[5].partition(":")
wm, wn = int(wm), (int(wn) if wn else 0)
secs = float(sys.argv[6]) if len(sys.argv) > 6 else 4.0
dev = torch.device("cuda:0"); lib = _lib.load()
taps = 9 if wm not in (1, 8) else 1
x = View(torch.randn((8, h, w, cin), device=dev)); pc = ops.PackedConv(torch.randn((cout, cin, 3 if taps == 9 else 1, 3 if taps == 9 else 1)) * 0.05, None, None, dev)
y = View(torch.empty((8, h, w, cout), device=dev))
d = (_lib.ConvDesc * 1)(); ops._fill_desc(d[0], x, pc, y, True, None, None, False, False)
if taps == 1 and wn: d[0].tune_wm, d[0].tune_sc, d[0].tune_wn = wm, 32, wn
if taps == 9: d[0].tune_wm, d[0].tune_sc, d[0].tune_wn = wm, 16, (1 if wm == 6 else 2)
samples = []
stop = False
def sampler():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "-d", "0"], capture_output=True, text=True, timeout=5).stdout
            pw = re.search(r"Power \(W\):\s*([\d.]+)", out) or re.search(r"Socket Power.*?:\s*([\d.]+)", out)
            sc = re.search(r"sclk clock level:?\s*\d*:?\s*\(?(\d+)Mhz", out)
            samples.append((time.time(), float(pw.group(1)) if pw else None, int(sc.group(1)) if sc else None, out if len(samples) == 0 else None))
        except Exception as e:
            samples.append((time.time(), None, None, str(e)))
        time.sleep(0.25)
for _ in range(5): lib.cmk_conv2d_nhwc(ctypes.byref(d[0]), ops._stream())
torch.cuda.synchronize()
th = threading.Thread(target=sampler); th.start()
t0 = time.time(); n = 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
while time.time() - t0 < secs:
    for _ in range(50): lib.cmk_conv2d_nhwc(ctypes.byref(d[0]), ops._stream())
    n += 50
    torch.cuda.synchronize()
e1.record(); torch.cuda.synchronize()
stop = True; th.join()
ms = e0.elapsed_time(e1) / n
fl = 2.0 * 8 * h * w * cin * cout * taps
print("%dx%d %d->%d wm %d: %.3f ms per launch over %.1f s, %.1f alg TFLOP/s" % (h, w, cin, cout, wm, ms, secs, fl / ms / 1e9))
print("rocm-smi samples (power W, sclk MHz):", [(p, c) for _, p, c, _ in samples][:24])
if samples and samples[0][3]: print(samples[0][3][:1500])
