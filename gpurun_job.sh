cd $GRAFT_REPO_ROOT
python tools/kbench.py --iters 100 --check 2>&1 | grep -v "^Loading\|^OBJ\|^Loaded\|^scene"
timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
