import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from centermask2_amd import ops, _lib, synthetic as S
from centermask2_amd.ops import View
from tests.helpers import build_gpu_model
model, sd = build_gpu_model("V-99-eSE")
x = S.make_synthetic_images(1, 128, 192, seed0=555)
with torch.no_grad():
    out = model.inference_padded(x.cuda(), [(128, 192)], want=("roi_feat",))
roi = out["roi_feat"]          # (50,14,14,256)
print("roi", tuple(roi.shape), float(roi.abs().max()), float(roi.abs().mean()), "zeros frac", float((roi == 0).float().mean()))
lib = _lib.load()
xin = View(roi.contiguous())
for k in range(1, 5):
    w = sd["roi_heads.mask_head.mask_fcn%d.weight" % k]; b = sd["roi_heads.mask_head.mask_fcn%d.bias" % k]
    pc = ops.PackedConv(w, None, b, "cuda")
    ref = F.relu(F.conv2d(xin.t.permute(0, 3, 1, 2).cpu().double(), w.double(), b.double(), padding=1)).float()
    res = {}
    for name, tv in (("direct", (2, 16, 4)), ("wino", (5, 16, 2))):
        y = View(torch.empty((50, 14, 14, 256), device="cuda"))
        d = (_lib.ConvDesc * 1)(); ops._fill_desc(d[0], xin, pc, y, True, None, None, False, False)
        d[0].tune_wm, d[0].tune_sc, d[0].tune_wn = tv
        assert lib.cmk_conv2d_nhwc(ctypes.byref(d[0]), ops._stream()) == 0
        torch.cuda.synchronize()
        got = y.t.permute(0, 3, 1, 2).cpu()
        err = (got - ref).abs()
        res[name] = y
        print("layer", k, name, "max ref %.3g max err %.3g rms rel %.2e  worst idx %s" % (ref.abs().max(), err.max(), err.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt(), tuple(int(v) for v in torch.nonzero(err == err.max())[0])))
    xin = res["direct"]
