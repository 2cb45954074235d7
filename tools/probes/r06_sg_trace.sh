cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/r06_sg_prof
export SG_MANIFEST=$GRAFT_REPO_ROOT/gpurun_out/r06_sg_manifest.json
rocprofv3 --kernel-trace -d gpurun_out/r06_sg_prof --output-format csv -- python3 tools/probes/sg_probe.py "$@" > gpurun_out/r06_sg_probe.txt 2> gpurun_out/r06_sg_probe.err
python3 tools/probes/sg_trace.py gpurun_out/r06_sg_prof gpurun_out/r06_sg_manifest.json > gpurun_out/r06_sg_trace.txt 2>&1
tail -5 gpurun_out/r06_sg_trace.txt
rm -rf gpurun_out/r06_sg_prof
