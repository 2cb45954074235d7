# Counters of wino_input_h2_kernel on the p2 layer (the transform runs at 4.6-4.8 TB/s and four store layouts tied
# without anyone looking at its TCP / TCC counters).  Separate --pmc passes; sums over all XCDs / instances, last dispatch.
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
mkdir -p gpurun_out
i=0
for C in "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" \
         "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum" "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WR_UNCACHED_32B_sum" "TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TCP_TCC_READ_REQ_LATENCY_sum" \
         "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES"; do
  i=$((i+1)); rm -rf /tmp/pw$i
  rocprofv3 --pmc $C -d /tmp/pw$i --output-format csv -- python3 tools/wino_one.py 64x120x160x256x256 > /tmp/pw$i.log 2>&1
  echo "== $C"
  python3 tools/pmc_dump.py /tmp/pw$i wino_input_h2 -1 | tail -n +2
done
grep wino_input /tmp/pw1.log | cut -c1-120
