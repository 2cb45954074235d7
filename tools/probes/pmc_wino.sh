cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
F=${1:-32}
i=0
for C in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" "SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA" "SQ_INSTS_LDS SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_WAVES"; do
  i=$((i+1)); rm -rf /tmp/pw$i
  timeout 120 rocprofv3 --pmc $C -d /tmp/pw$i --output-format csv -- python3 tools/wino_one.py ${F}x120x160x256x256 > /tmp/pw$i.log 2>&1 </dev/null
  echo "== wino [$C]"; python3 tools/pmc_dump.py /tmp/pw$i wino_gemm | head -4
done
