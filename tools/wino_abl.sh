#!/bin/bash
# Dev tool (GPU box): timing-only ablations of the fp16x2 Winograd GEMM's ring loop (results are NOT valid with a bit set).
#   A3D_WINO_ABL bits: 1 no V DMA, 2 no filter DMA, 4 no barrier, 8 no fragment reads, 16 no fold
A3D_HIPCC_FLAGS=-DA3D_ABLATIONS python -m articulation3d_amd.build > /dev/null || exit 1
for t in ${ABLS:-0 1 2 3 7 11 15 19 31}; do
  echo "A3D_WINO_ABL=$t: $(A3D_WINO_ABL=$t python tools/wino_one.py ${SHAPES:-64x120x160x256x256} 2>&1 | grep -v amdgpu | cut -c1-120)"
done
