"""Time cmk_nms_topk on synthetic candidates: python tools/bench_nms.py [candidates per image]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centermask2_amd import ops
dev = torch.device("cuda:0")
m = int(sys.argv[1]) if len(sys.argv) > 1 else 16000
n, cap = 8, 32768
g = torch.Generator().manual_seed(3)
xy = torch.rand((n, m, 2), generator=g) * torch.tensor([1280.0, 800.0])
wh = torch.rand((n, m, 2), generator=g) * 200 + 8
cand = dict(box=torch.zeros((n, cap, 4)), score=torch.zeros((n, cap)), cls=torch.zeros((n, cap), dtype=torch.int32),
            loc=torch.zeros((n, cap, 2)), counts=torch.full((n,), m, dtype=torch.int32))
cand["box"][:, :m] = torch.cat([xy, xy + wh], 2); cand["score"][:, :m] = torch.rand((n, m), generator=g); cand["cls"][:, :m] = torch.randint(0, 80, (n, m), generator=g).int()
cand = {k: v.to(dev) for k, v in cand.items()}
for topk in (50, 100):
    for _ in range(3): ops.nms_topk(cand, 0.6, topk)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.nms_topk(cand, 0.6, topk)
    e1.record(); torch.cuda.synchronize()
    print("nms_topk %d candidates x %d images, topk %d: %.1f us" % (m, n, topk, e0.elapsed_time(e1) / 20 * 1e3))
