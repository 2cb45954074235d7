# duty cycle and clock of a bare bf16 MFMA loop: kernel durations (trace) + MFMA-busy / GUI-active cycles (PMC pass)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT/tools/probes"
hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_shape_bf16 mfma_shape_bf16.hip || exit 1
/tmp/mfma_shape_bf16
rm -rf /tmp/mp1 /tmp/mp2
rocprofv3 --kernel-trace --stats -d /tmp/mp1 --output-format csv -- /tmp/mfma_shape_bf16 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/mp1/**/*kernel_stats.csv", recursive=True)[0]
print(open(f).read()[:1500])
PY
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -d /tmp/mp2 --output-format csv -- /tmp/mfma_shape_bf16 > /dev/null 2>&1
python3 "$GRAFT_REPO_ROOT/tools/pmc_dump.py" /tmp/mp2 loop
