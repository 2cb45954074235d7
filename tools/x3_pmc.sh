# PMC passes behind the bf16x3 numbers of DESIGN.md section 5 (MFMA busy, LDS bank conflicts, clock = GRBM_GUI_ACTIVE / 8 / wall): run through gpurun.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/x3pmc; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $O/a -- python3 tools/x3_one.py 2 > $O/a.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS --output-format csv -d $O/b -- python3 tools/x3_one.py 2 > $O/b.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d $O/c -- python3 tools/x3_one.py 2 > $O/c.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_LEVEL_WAVES SQ_ACTIVE_INST_VMEM --output-format csv -d $O/d -- python3 tools/x3_one.py 2 > $O/d.log 2>&1
python tools/x3_pmc_dump.py $O/a $O/b $O/c $O/d
tail -3 $O/*.log | cut -c1-300
