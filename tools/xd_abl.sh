#!/bin/bash
# Dev tool (GPU box): timing-only ablations of the dual-DMA ring of conv_h2w_kernel xd on fc1 (results are NOT valid with a bit set).
#   bit 10 (1024) no activation DMA, 11 (2048) no filter DMA, 12 (4096) no barrier
set -e
A3D_HIPCC_FLAGS=-DA3D_ABLATIONS python -m articulation3d_amd.build > /dev/null
for t in 0 1024 2048 3072 4096 7168; do
  echo "== A3D_XD_TUNE=$t"
  A3D_XD_TUNE=$t python tools/fc1_bench.py 2>&1 | grep "fc1\|pool"
done
