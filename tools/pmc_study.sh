# Stall / issue counters of the two dominant kernel families on one layer each (separate --pmc passes, no tracing):
#   wino_gemm_h2w (p2 3x3 256 -> 256) and conv_h2<2> (res4 1x1 256 -> 1024 + residual).   bash tools/pmc_study.sh [frames]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
F=${1:-32}
i=0
for C in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" "SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA" "SQ_INSTS_LDS SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM"; do
  i=$((i+1)); rm -rf /tmp/pw$i /tmp/px$i
  rocprofv3 --pmc $C -d /tmp/pw$i --output-format csv -- python3 tools/wino_one.py ${F}x120x160x256x256 > /tmp/pw$i.log 2>&1
  echo "== wino [$C]"; python3 tools/pmc_dump.py /tmp/pw$i wino_gemm | head -6
  rocprofv3 --pmc $C -d /tmp/px$i --output-format csv -- python3 tools/x3_tile_ab.py ${F}x30x40x256x1024x1x1x1 > /tmp/px$i.log 2>&1
  echo "== conv_h2 [$C]"; python3 tools/pmc_dump.py /tmp/px$i conv_x3_kernel | head -12
done
