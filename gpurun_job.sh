cd $GRAFT_REPO_ROOT
for b in 1 8; do python tools/kbench.py --iters 50 --batch $b --check 2>&1 | grep -v "^Loading\|^OBJ\|^Loaded\|^scene"; done
python tools/kbench.py --scene atrium --width 3840 --height 2160 --iters 20 --batch 4 --check 2>&1 | grep -v "^Loading\|^OBJ\|^Loaded\|^scene"
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
