cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_SALU SQ_BUSY_CYCLES" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES"; do
  i=$((i+1)); rm -rf /tmp/ps$i
  timeout 120 rocprofv3 --pmc $C -d /tmp/ps$i --output-format csv -- python3 tools/probes/stem_ab.py > /tmp/ps$i.log 2>&1 </dev/null
  echo "== [$C]"; python3 tools/pmc_dump.py /tmp/ps$i conv_xs_stem | head -6
done
