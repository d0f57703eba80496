#!/bin/bash
# Builds a SECOND copy of the library with ONE translation unit replaced (coach_amd/ab/librlx_<name>.so, not used by the
# package) so that tools/ab_c2_pipeline.py / tools/ppo_update_once.py --lib can time two builds on one box.
#   bash tools/ab_lib.sh depth4 coach_amd/csrc/gemm.hip -DRLX_GEMM_DEPTH=4
#   git show HEAD~1:coach_amd/csrc/gemm.hip > /tmp/gemm_prev.hip && bash tools/ab_lib.sh prev /tmp/gemm_prev.hip
#   UNIT=ppo_heads_bwd bash tools/ab_lib.sh prev /tmp/heads_prev.hip          (replaces build/ppo_heads_bwd.o; default UNIT=gemm)
set -eu
NAME=$1; SRC=$(readlink -f $2); shift 2
UNIT=${UNIT:-gemm}
cd "$(dirname "$0")/../coach_amd/csrc"
mkdir -p ../ab build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -I../../include -I. \
    "$@" -c $SRC -o build/ab_${UNIT}_$NAME.o
objs=$(ls build/*.o | grep -v "build/ab_" | grep -v "build/$UNIT.o" | grep -v "gemm_depth")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../ab/librlx_$NAME.so $objs build/ab_${UNIT}_$NAME.o
ls -la ../ab/librlx_$NAME.so
