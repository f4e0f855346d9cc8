cd $GRAFT_REPO_ROOT
python -c "
import importlib,sys; sys.path.insert(0,'.')
rt=importlib.import_module('cuda-raytracing_amd'); rt.build()"
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "8k" 2>&1 | tail -5
