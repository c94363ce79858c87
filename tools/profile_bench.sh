#!/bin/bash
# rocprofv3 passes of the bench command, as the MI355X guide prescribes: --kernel-trace --stats in one run, every --pmc set in a run of
# its own (never combined with --sys-trace / runtime traces).  Writes raw output under gpurun_out/prof_<tag>/ and the summaries that are
# judged under profiles/<tag>_*.   usage (on the GPU box, from the repo root): tools/profile_bench.sh r02 [bench args...]
set -e
TAG=${1:-r02}; shift || true
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT $ROOT/profiles
export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-extras $*"
cd /tmp
echo "== kernel trace + stats"; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python3 $ROOT/bench.py $ARGS > $OUT/bench_under_stats.json 2> $OUT/stats.err || tail -3 $OUT/stats.err
echo "== pmc mfma";  rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -d $OUT/mfma -o m -- python3 $ROOT/bench.py $ARGS > $OUT/bench_under_mfma.json 2> $OUT/mfma.err || tail -3 $OUT/mfma.err
echo "== pmc fetch"; rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/fetch -o f -- python3 $ROOT/bench.py $ARGS > /dev/null 2> $OUT/fetch.err || tail -3 $OUT/fetch.err
echo "== pmc write"; rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/write -o w -- python3 $ROOT/bench.py $ARGS > /dev/null 2> $OUT/write.err || tail -3 $OUT/write.err
cd $ROOT
find $OUT -name "*.csv" | head -20
ST=$(find $OUT/stats -name "*kernel_stats.csv" | head -1); [ -n "$ST" ] && head -40 $ST > profiles/${TAG}_bench_kernel_stats.csv
cp $OUT/bench_under_stats.json profiles/${TAG}_bench_under_rocprof.json || true
python3 tools/pmc_mfma.py $(find $OUT/mfma -name "*counter_collection.csv" | head -1) profiles/${TAG}_pmc_mfma.json
python3 tools/pmc_traffic.py $(find $OUT/fetch -name "*counter_collection.csv" | head -1) $(find $OUT/write -name "*counter_collection.csv" | head -1) profiles/${TAG}_pmc_traffic.json
