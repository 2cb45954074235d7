cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/r06_bsg_prof
export SG_MANIFEST=$GRAFT_REPO_ROOT/gpurun_out/r06_bsg_manifest.json
rocprofv3 --kernel-trace -d gpurun_out/r06_bsg_prof --output-format csv -- python3 tools/probes/bsg_probe.py "$@" > gpurun_out/r06_bsg_probe.txt 2> gpurun_out/r06_bsg_probe.err
python3 tools/probes/sg_trace.py gpurun_out/r06_bsg_prof gpurun_out/r06_bsg_manifest.json > gpurun_out/r06_bsg_trace.txt 2>&1
tail -5 gpurun_out/r06_bsg_trace.txt; tail -3 gpurun_out/r06_bsg_probe.err
rm -rf gpurun_out/r06_bsg_prof
