"""One FCOS tower conv (256 -> 256, 5 FPN levels of 8 x 3x800x1280 in one launch, GroupNorm+ReLU of the producer fused into the input staging,
GroupNorm statistics of the output from the epilogue) on the two F(4x4) kernels: tune (6,16,1) vs (6,64,1).  python tools/bench_fcos_tower.py [rounds]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centermask2_amd import ops
dev = torch.device("cuda:0"); B = 8
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
shapes = [(100, 160), (50, 80), (25, 40), (13, 20), (7, 10)]
g = torch.Generator().manual_seed(3)
xs = [ops.View(torch.randn((B, h, w, 256), generator=g).to(dev)) for h, w in shapes]
pc = ops.PackedConv(torch.randn((256, 256, 3, 3), generator=g) * 0.02, None, torch.zeros(256), dev)
gamma, beta = torch.ones(256, device=dev), torch.zeros(256, device=dev)
aff = [(torch.rand((B, 256), generator=g).to(dev) + 0.5, torch.randn((B, 256), generator=g).to(dev) * 0.1) for _ in shapes]
def run(tv, with_aff):
    ops.FORCE_VARIANT = tv
    return ops.conv_gn_multi(xs, [pc] * 5, gamma, beta, 32, 1e-5, in_affine=aff if with_aff else None)
def timeit(tv, with_aff, it=10):
    run(tv, with_aff); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): run(tv, with_aff)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
for with_aff in (False, True):
    best = {}
    for _ in range(rounds):
        for tv in ((6, 16, 1), (6, 64, 1)):
            best[tv] = min(best.get(tv, 1e9), timeit(tv, with_aff))
    ya = run((6, 16, 1), with_aff)[0]; yb = run((6, 64, 1), with_aff)[0]; torch.cuda.synchronize()
    print("fused input affine %-5s  w6 %.3f ms   w6s %.3f ms   (conv + GN finalize; equal outputs: %s)" % (with_aff, best[(6, 16, 1)], best[(6, 64, 1)],
          all(torch.equal(a.t, b.t) for a, b in zip(ya, yb))), flush=True)
