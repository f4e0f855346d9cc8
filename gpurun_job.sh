cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330
for b in 8 20 25; do python tools/kbench.py --iters 20 --batch $b 2>&1 | grep "^mid"; done
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29515 bench.py --gpus 4 --steps 50 --warmup 10 --debug-backend gloo 2>&1 | grep '^{"metric"' | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('rehearsal N=4: frame ok', d['frame_matches_debug_kernel'], 'F', d['config']['frames_per_launch'])"
