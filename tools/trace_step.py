"""Timeline of ONE graph replay from a rocprofv3 --kernel-trace CSV: kernel sequence with start offsets, durations and the idle gap in
front of each kernel; totals of busy time, gaps and overlap.  usage: trace_step.py <kernel_trace.csv> [n_last_kernels_per_step]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# one step = the span between two consecutive launches of the stem kernel (the first kernel of the model)
stem = [i for i, n in enumerate(names) if "stem_conv_kernel<1>" in n or "preprocess" in n]
if len(stem) < 2:
    stem = [i for i, n in enumerate(names) if "stem_conv_kernel" in n][::2]
i0, i1 = stem[-2], stem[-1]
step = rows[i0:i1]
t0 = int(step[0]["Start_Timestamp"])
end_prev = t0
busy = 0; gaps = 0; gap_by = collections.Counter(); dur_by = collections.Counter(); cnt_by = collections.Counter()
print("kernels in the step:", len(step), "span %.3f ms" % ((int(rows[i1]["Start_Timestamp"]) - t0) / 1e6))
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = s - end_prev
    nm = r["Kernel_Name"].replace("void ", "").replace("cmk::", "")[:60]
    if gap > 0: gaps += gap; gap_by[nm] += gap
    busy += e - max(s, end_prev) if e > end_prev else 0
    dur_by[nm] += e - s; cnt_by[nm] += 1
    print("%9.1f us  dur %8.1f  gap %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap / 1e3, nm))
    end_prev = max(end_prev, e)
print("busy %.3f ms, gaps %.3f ms" % (busy / 1e6, gaps / 1e6))
print("gap in front of (top):")
for k, v in gap_by.most_common(15): print("  %8.1f us  x%-4d %s" % (v / 1e3, cnt_by[k], k))
print("duration by kernel (top):")
for k, v in dur_by.most_common(25): print("  %8.1f us  x%-4d %s" % (v / 1e3, cnt_by[k], k))
