cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -q -m gpu -x -k "bn_head" 2>&1 | tail -3
timeout 600 python tools/bench_head.py 2>&1 | grep -v amdgpu.ids
