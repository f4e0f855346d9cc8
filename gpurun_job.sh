cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/trace
RT_TRACE_PROF=1 python tools/trace_one.py gpurun_out/trace 2>&1 | grep -v "^Loading\|^OBJ\|^Loaded" | head -20
rm -f gpurun_out/trace/*.bin
