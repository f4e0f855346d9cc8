cd $GRAFT_REPO_ROOT
python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330
python bench.py --no-cpu-baseline --camera far 2>&1 | tail -1 | cut -c1-330
