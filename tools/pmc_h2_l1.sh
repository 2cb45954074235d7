# L1 -> L2 request counters of the fp16x2 narrow direct kernel on two 1x1 layers (is each 128-byte line of X fetched once or twice?)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for C in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TCP_TCC_READ_REQ_LATENCY_sum"; do
  i=$((i+1)); rm -rf /tmp/pl$i
  X3_TUNES=0 rocprofv3 --pmc $C -d /tmp/pl$i --output-format csv -- python3 tools/x3_tile_ab.py 64x30x40x256x1024x1x1x1 64x120x160x64x256x1x1x1 64x30x40x1024x256x1x1x0 > /tmp/pl$i.log 2>&1
  echo "== $C"
  tail -3 /tmp/pl$i.log | cut -c1-150
  python3 tools/pmc_dump.py /tmp/pl$i conv_x3_kernel -2
done
