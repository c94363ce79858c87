#!/bin/bash
# End-to-end A/B of library builds: tools/ab/ab_bench.sh <rounds> <lib.so> [<lib.so> ...] — bench.py per build, interleaved; prints img/s and the
# per-step milliseconds of the conv kernels (HIP events) for each run.
R=$1; shift
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
for r in $(seq 1 $R); do
  for L in "$@"; do
    CMK_LIB=$(readlink -f $L) python3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys
d=json.load(sys.stdin); pk=d['roofline']['per_kernel']
print('%-28s %7.2f img/s  ' % ('$(basename $L)', d['value']) + '  '.join('%s %.2f' % (k.replace('conv_','').replace('_kernel',''), v['ms']) for k,v in list(pk.items())[:5]))"
  done
done
