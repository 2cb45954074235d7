# stall counters of the fp16x2 narrow direct kernel on one compute-bound layer (res4 1x1 1024 -> 256, 64 frames)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for C in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" "SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA"; do
  i=$((i+1)); rm -rf /tmp/ph$i
  rocprofv3 --pmc $C -d /tmp/ph$i --output-format csv -- python3 tools/x3_tile_ab.py 64x30x40x1024x256x1x1x0 > /tmp/ph$i.log 2>&1
  python3 tools/pmc_dump.py /tmp/ph$i conv_x3_kernel | grep -A5 "2, false, true" | head -6
done
