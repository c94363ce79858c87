#!/bin/bash
# build_variant.sh <name> <conv_wino6 | conv_pw source> [extra hipcc flags]  ->  centermask2_amd/ab/libcmk_<name>.so
# (same objects as libcmk_hip.so except the object of that source (its file name starts with conv_wino6 or conv_pw): for same-session
# A/B runs of kernel variants, tools/ab/ab_wino6.py, tools/bench_pw.py under CMK_LIB)
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
NAME=$1; SRC=$(readlink -f "$2"); shift 2
cd "$ROOT/centermask2_amd/csrc"
mkdir -p ../ab build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I. -c -x hip "$SRC" -o build/w6_$NAME.o "$@"
case "$(basename "$SRC")" in conv_pw*) REPL=conv_pw.o;; conv_wino6s*) REPL=conv_wino6s.o;; conv_sp3*) REPL=conv_sp3.o;; *) REPL=conv_wino6.o;; esac
OBJS=$(ls build/*.o | grep -v "build/w6_" | grep -v $REPL)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../ab/libcmk_$NAME.so $OBJS build/w6_$NAME.o
echo "built centermask2_amd/ab/libcmk_$NAME.so"
