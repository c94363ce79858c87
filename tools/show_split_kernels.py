import json
d=json.loads(open("gpurun_out/bench_sk.json").read().strip().split(chr(10))[-1])
print(d["value"], d["split_direct3x3"]["images_per_sec"], d["split_direct3x3"]["all_convs_ms_instrumented"])
for k,v in d["split_direct3x3"]["split_kernels"].items(): print(k, v)
