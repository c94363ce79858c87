"""Matrix-pipe utilisation per kernel from one rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE pass
(counter_collection.csv; its own run, --kernel-trace only, as the MI355X guide prescribes).
  mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES)      per dispatch, averaged over the kernel's dispatches
  clock_ghz      = GRBM_GUI_ACTIVE / 8 XCDs / dispatch duration                  (reads high on dispatches under ~0.3 ms)
Dispatches whose busy counter sits on a multiple of 2^20 * 1000 are flagged: round 1 saw that value for two different kernels
(a clipped counter), so such samples are excluded from the average.
usage: pmc_mfma.py <counter_collection.csv> <out.json> [<kernel_trace.csv>]"""
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from centermask2_amd.ops import kernel_source_hash

rows = collections.defaultdict(dict)
names = {}
dur_csv = {}
for r in csv.DictReader(open(sys.argv[1])):
    did = r.get("Dispatch_Id") or r.get("Dispatch_ID")
    rows[did][r["Counter_Name"]] = float(r["Counter_Value"])
    names[did] = r["Kernel_Name"].replace("void cmk::", "").replace("(cmk::ConvArgs)", "")
    if r.get("Start_Timestamp") and r.get("End_Timestamp"):
        dur_csv[did] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
dur = dict(dur_csv)
if len(sys.argv) > 3:
    for r in csv.DictReader(open(sys.argv[3])):
        dur[r.get("Dispatch_Id") or r.get("Dispatch_ID")] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
acc = collections.defaultdict(lambda: dict(busy=[], clipped=0, clk=[]))
for did, c in rows.items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" not in c or not c.get("SQ_BUSY_CU_CYCLES"):
        continue
    a = acc[names[did]]
    if c["SQ_VALU_MFMA_BUSY_CYCLES"] > 0 and c["SQ_VALU_MFMA_BUSY_CYCLES"] % 1048576000.0 == 0.0:
        a["clipped"] += 1
        continue
    a["busy"].append(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * c["SQ_BUSY_CU_CYCLES"]))
    if did in dur and dur[did] > 0 and "GRBM_GUI_ACTIVE" in c:
        a["clk"].append(c["GRBM_GUI_ACTIVE"] / 8.0 / dur[did])
out = {"_kernel_source_hash": kernel_source_hash()}
for k, a in acc.items():
    if a["busy"]:
        out[k] = {"mfma_busy_frac": sum(a["busy"]) / len(a["busy"]), "dispatches": len(a["busy"]), "clipped_dispatches": a["clipped"],
                  "clock_ghz": (sum(a["clk"]) / len(a["clk"])) if a["clk"] else None}
json.dump(out, open(sys.argv[2], "w"), indent=1, sort_keys=True)
print("wrote", sys.argv[2], len(out) - 1, "kernels")
