#!/bin/bash
# build_variant.sh <name> <conv_wino6 source> [extra hipcc flags]  ->  centermask2_amd/ab/libcmk_<name>.so
# (same objects as libcmk_hip.so except conv_wino6.o: for same-session A/B runs of kernel variants, tools/ab/ab_wino6.py)
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
NAME=$1; SRC=$(readlink -f "$2"); shift 2
cd "$ROOT/centermask2_amd/csrc"
mkdir -p ../ab build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I. -c -x hip "$SRC" -o build/w6_$NAME.o "$@"
OBJS=$(ls build/*.o | grep -v "build/w6_" | grep -v conv_wino6.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../ab/libcmk_$NAME.so $OBJS build/w6_$NAME.o
echo "built centermask2_amd/ab/libcmk_$NAME.so"
