#!/bin/bash
# tools/ab/ab_pw.sh <rounds> <lib.so> [<lib.so> ...]: tools/bench_pw.py under each library build, back to back on one box
R=$1; shift
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
for L in "$@"; do
  echo "== $(basename $L)"
  CMK_LIB=$(readlink -f $L) python3 $ROOT/tools/bench_pw.py $R 2>/dev/null
done
