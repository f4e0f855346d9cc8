cd $GRAFT_REPO_ROOT
echo "== baseline"; python tools/kbench.py --iters 40 --batch 8 2>&1 | grep -v "^Loading\|^OBJ\|^Loaded\|^scene"
for w in 2 6; do cp cuda-raytracing_amd/librt_hip_vote$w.so cuda-raytracing_amd/librt_hip.so; touch cuda-raytracing_amd/librt_hip.so cuda-raytracing_amd/librt_host.so
echo "== vote wait $w"; python tools/kbench.py --iters 40 --batch 8 --check 2>&1 | grep -v "^Loading\|^OBJ\|^Loaded\|^scene"; done
