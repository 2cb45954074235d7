#!/bin/bash
# Dev tool (GPU box): timing-only ablations of the ping-pong Winograd GEMM loop (results are NOT valid with a bit set).
#   A3D_WINO_ABL bits: 1 no V DMA, 2 no filter DMA, 8 no fragment reads, 16 no fold, 32 no MFMAs
A3D_HIPCC_FLAGS=-DA3D_ABLATIONS python3 -m articulation3d_amd.build > /dev/null || exit 1
for t in ${ABLS:-0 1 2 3 8 11 32 35 40 43}; do
  echo "A3D_WINO_ABL=$t: $(A3D_WINO_ABL=$t TUNES=${TUNES:-0,21} python3 tools/wino_pp_ab.py ${SHAPES:-64x120x160x256x256} 2>&1 | grep -v amdgpu | cut -c1-200)"
done
