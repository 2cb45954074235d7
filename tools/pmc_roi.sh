# L1 / L2 request counters of the ROI pooler (tools/roi_bench.py runs the three poolers on the detector's own boxes)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for C in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_BUSY_avr" "TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD"; do
  i=$((i+1)); rm -rf /tmp/pr$i
  rocprofv3 --pmc $C -d /tmp/pr$i --output-format csv -- python3 tools/roi_bench.py --frames 64 > /tmp/pr$i.log 2>&1
  echo "== $C"; tail -2 /tmp/pr$i.log | cut -c1-200
  python3 tools/pmc_dump.py /tmp/pr$i roi_align_fpn_kernel 38 | head -12
done
