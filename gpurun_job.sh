cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/trace
python tools/trace_one.py gpurun_out/trace 2>&1 | grep -v "^Loading\|^OBJ\|^Loaded" | head -12
RT_TRACE_PROF=1 python tools/trace_one.py gpurun_out/trace 2>&1 | grep -v "^Loading\|^OBJ\|^Loaded" | grep "all waves"
rm -f gpurun_out/trace/*.bin
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
