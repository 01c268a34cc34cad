#!/bin/bash
# A/B library of the ping-pong halo kernel: tools/build_h8_variant.sh <name> [-DPP_H8_PRIO=0 -DPP_H8_DMA_IN_MFMA=0 ...]
# compiles only the two translation units that instantiate conv_halo8.h with the extra flags and links them with the cached objects
# of the production library (propainter_amd/lib/obj) -> build/<name>/libpropainter_hip.so (LD_LIBRARY_PATH / PP_LIB_PATH select it).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
OUT=$R/build/$NAME
mkdir -p $OUT
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result"
/opt/rocm/bin/hipcc $F "$@" -c $R/propainter_amd/csrc/conv_gemm_v3.hip -o $OUT/conv_gemm_v3.o 2>/dev/null &
/opt/rocm/bin/hipcc $F "$@" -c $R/propainter_amd/csrc/conv_gemm_v3s.hip -o $OUT/conv_gemm_v3s.o 2>/dev/null &
wait
OTHERS=$(ls $R/propainter_amd/lib/obj/*.o | grep -v "conv_gemm_v3\.\|conv_gemm_v3s\.")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libpropainter_hip.so $OUT/conv_gemm_v3.o $OUT/conv_gemm_v3s.o $OTHERS
rm -f $OUT/*.o
echo built $OUT/libpropainter_hip.so
