"""Five eager steps of the opt-in fp16-split path (tuned/*_split3.json) for a kernel trace:
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_split3 -o s -- python3 tools/prof_split3.py [body]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centermask2_amd import ops, synthetic as S
import bench
body = sys.argv[1] if len(sys.argv) > 1 else "V-39-eSE"
ops.ALLOW_SPLIT_BF16 = ops.ALLOW_SPLIT_F16 = True
ops._TUNED.clear()
ops.load_tuned(os.path.join(bench.ROOT, "centermask2_amd", "tuned", "mi355x_{}_b8_800x1280_split3.json".format(body)))
dev = torch.device("cuda:0")
model, _ = bench.build(body, dev)
x = S.make_synthetic_images(8, 800, 1280, seed0=1234).to(dev)
with torch.no_grad():
    for _ in range(6):
        model.inference_padded(x, [(800, 1280)] * 8)
torch.cuda.synchronize()
