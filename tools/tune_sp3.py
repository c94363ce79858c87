"""Targeted re-tuning of the 3x3 stride-1 convs for the opt-in direct bf16-split form (conv_sp3.hip, tune 11 / 2 / geometry): every such
problem of the model is timed on its real buffers with the shipped table's choice and the four geometries, interleaved, and a table with
the winners is written.  python tools/tune_sp3.py <body> <out.json> [margin]   (a geometry must beat the incumbent by `margin`, default 1.03)"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centermask2_amd import ops
import bench
body, out = sys.argv[1], sys.argv[2]
margin = float(sys.argv[3]) if len(sys.argv) > 3 else 1.03
B = 8
dev = torch.device("cuda:0")
ops.ALLOW_SPLIT_BF16 = ops.ALLOW_SPLIT_F16 = True
ops._TUNED.clear()
ops.load_tuned(os.path.join(bench.ROOT, "centermask2_amd", "tuned", "mi355x_{}_b{}_800x1280.json".format(body, B)))
moved = 0
for k, v in list(ops._TUNED.items()):            # every conv the pointwise GEMM kernel runs without split-K: its fp16 two-piece form (tune 12)
    if len(v) == 3 and ((k[0] == 1 and v[0] == 8 and k[6] in (0, 2)) or (k[0] == 3 and v[0] == 9 and k[6] == 0)):
        ops._TUNED[k] = (12, 32, 4); moved += 1
incumbent = {k: v for k, v in ops._TUNED.items() if k[0] == 3 and k[1] == 1}
for k in incumbent: del ops._TUNED[k]
ops.TUNE_ONLY = lambda key: [incumbent[key]] + [(11, 2, g) for g in range(4)] + [(11, 21, g) for g in range(4)] if key in incumbent else [(0, 0, 0)]
ops.TUNE_REPS, ops.TUNE_ROUNDS = 4, 3
ops.PAIR_TOWERS = False                          # the tower launches are tuned as single-tower launches (the pair takes that entry)
ops.set_autotune(True)
from centermask2_amd import synthetic as S
model, _ = bench.build(body, dev)
x = S.make_synthetic_images(B, 800, 1280, seed0=1234).to(dev)
with torch.no_grad():
    model.inference_padded(x, [(800, 1280)] * B)
torch.cuda.synchronize()
won = 0
for key, times in ops.TUNE_LOG:
    if key not in incumbent: continue
    inc = tuple(incumbent[key]) + (1,) * (4 - len(incumbent[key]))
    t_inc = times.get(inc, float("inf"))
    best = min(times, key=times.get)
    keep = best if (best[0] != 11 or times[best] * margin < t_inc) else inc
    ops._TUNED[key] = keep[:3] if keep[3] == 1 else keep
    won += keep[0] == 11
    print("%-90s inc %s %.3f | %s | -> %s" % (ops._key_to_str(key), incumbent[key], t_inc, " ".join("g%d%s %.3f" % (tv[2], "n" if tv[1] == 21 else "", ms) for tv, ms in times.items() if tv[0] == 11), ops._TUNED[key]), flush=True)
ops.save_tuned(out)
print("3x3 problems: %d, moved to the direct split form: %d; pointwise convs on the fp16 two-piece form: %d; table: %s" % (len(incumbent), won, moved, out))
