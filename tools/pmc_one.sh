#!/bin/bash
# PMC passes of ONE conv shape with a forced variant: tools/pmc_one.sh <tag> H W Cin Cout wm sc wn  (on the GPU box, from the repo root)
# PMC_SETS="A B;C D" replaces the default counter passes.
# (counters in their own runs, --kernel-trace only, as the MI355X guide prescribes)
TAG=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd); OUT=$ROOT/gpurun_out/pmc_$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
i=0
if [ -n "$PMC_SETS" ]; then IFS=';' read -ra SETS <<< "$PMC_SETS"; else SETS=(); fi
for SET in "${SETS[@]:-}" "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_COEXEC_CYCLES"; do
  [ -z "$SET" ] && continue
  [ -n "$PMC_SETS" ] && [ $i -ge ${#SETS[@]} ] && break       # PMC_SETS="A B;C D": only those passes
  i=$((i+1))
  rocprofv3 --kernel-trace --output-format csv --pmc $SET -d $OUT/p$i -o c -- python3 $ROOT/tools/prof_wino.py "$@" 2 > /dev/null 2> $OUT/p$i.err || tail -2 $OUT/p$i.err
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/p*/c_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "conv_wino" in r["Kernel_Name"] or "conv_sp3" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc): print("%-32s %.4g  (n=%d)" % (k, sum(acc[k]) / len(acc[k]), len(acc[k])))
PY
rm -rf $OUT/p*/c_kernel_trace.csv
