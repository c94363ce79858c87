#!/bin/bash
# Build libcmk_hip.so for gfx950 in-tree (travels to the GPU box with the snapshot).
set -e
cd "$(dirname "$0")"
OUT=../libcmk_hip.so
SRCS=$(ls *.hip)
OBJS=""
mkdir -p build
for s in $SRCS; do
  o=build/${s%.hip}.o
  if [ ! -f "$o" ] || [ "$s" -nt "$o" ] || [ cmk_common.hpp -nt "$o" ] || [ conv_args.hpp -nt "$o" ] || [ wino6_common.hpp -nt "$o" ] || [ ../../include/cmk.h -nt "$o" ]; then
    echo "hipcc $s"
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c "$s" -o "$o" ${CMK_HIPCC_FLAGS}
  fi
  OBJS="$OBJS $o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT $OBJS
echo "built $OUT"
