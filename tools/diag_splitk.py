"""Debug helper: split-K / gather variants of one conv against torch."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from centermask2_amd import ops, _lib
from centermask2_amd.ops import View

def run(n, h, w, cin, cout, k, stride, tv, with_res):
    g = torch.Generator().manual_seed(1)
    x = torch.randn((n, cin, h, w), generator=g)
    wt = torch.randn((cout, cin, k, k), generator=g) * (2.0 / (cin * k * k)) ** 0.5
    ref = F.conv2d(x, wt, None, stride=stride, padding=k // 2)
    res = torch.randn(ref.shape, generator=g) if with_res else None
    if with_res: ref = ref + res
    ref = F.relu(ref)
    pc = ops.PackedConv(wt, None, None, "cuda", stride=stride); pc.w_wino = None
    y = View(torch.full((n, ref.shape[2], ref.shape[3], cout), -5.0, device="cuda"))
    d = (_lib.ConvDesc * 1)()
    xv, resv = ops.as_view(x.cuda()), (ops.as_view(res.cuda()) if with_res else None)     # keep alive: the descriptor borrows pointers
    ops._fill_desc(d[0], xv, pc, y, True, None, resv, False, False)
    ws = ops._set_variant(d, 1, tv)
    rc = _lib.load().cmk_conv2d_nhwc(ctypes.byref(d[0]), ops._stream())
    torch.cuda.synchronize()
    got = y.nchw().cpu()
    err = (got - ref).abs()
    unwritten = int((got == -5.0).sum())
    print(tv, "res" if with_res else "   ", "rc", rc, "max err %.3e" % err.max().item(), "unwritten", unwritten, "bad", int((err > 1e-3).sum()), "of", err.numel())
    if err.max() > 1e-3:
        idx = (err > 1e-3).nonzero()
        print("  first bad (n,c,h,w):", idx[:5].tolist(), "channels", sorted(set(idx[:, 1].tolist()))[:12])

for tv in [(0, 0, 0, 1), (0, 0, 0, 2), (1, 32, 1, 2), (1, 32, 3, 2), (2, 32, 1, 2), (2, 32, 3, 2)]:
    for r in (False, True):
        run(1, 1, 400, 1024, 80, 1, 1, tv, r)
