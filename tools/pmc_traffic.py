"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs as the MI355X guide prescribes) of the same
command into per-kernel HBM bytes per launch.  gfx950 corrections (MI355X_MICROARCH.md §HBM): both counters are in KiB;
FETCH_SIZE reports exactly half of the bytes of a wide (16 B/lane) coalesced read stream, so it is doubled; WRITE_SIZE is
exact for 16 B/lane stores.   usage: pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>"""
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"].replace("void cmk::", "").replace("(cmk::ConvArgs)", "")].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


fetch, nf = per_kernel(sys.argv[1], "FETCH_SIZE")
write, _ = per_kernel(sys.argv[2], "WRITE_SIZE")
out = {}
for k in fetch:
    if k.startswith("conv_") or k.startswith("cmk::"):
        out[k] = {"fetch_bytes_per_launch": 2.0 * 1024.0 * fetch[k], "write_bytes_per_launch": 1024.0 * write.get(k, 0.0),
                  "hbm_bytes_per_launch": 2.0 * 1024.0 * fetch[k] + 1024.0 * write.get(k, 0.0), "launches_sampled": nf[k],
                  "correction": "FETCH_SIZE KiB x1024 x2 (gfx950 half-count of wide reads), WRITE_SIZE KiB x1024"}
from centermask2_amd.ops import kernel_source_hash
out["_kernel_source_hash"] = kernel_source_hash()          # bench.py quotes a summary only while this matches the built sources
json.dump(out, open(sys.argv[3], "w"), indent=1, sort_keys=True)
print("wrote", sys.argv[3], len(out), "kernels")
