cd $GRAFT_REPO_ROOT
python tools/kbench.py --iters 40 --batch 8 --check 2>&1 | grep -v "^Loading\|^OBJ\|^Loaded\|^scene"
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
