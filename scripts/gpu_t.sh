set -e
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_enhance_large.py -m gpu -x -q -k "parity" 2>&1 | grep -v "^$" > gpurun_out/par.log || { tail -60 gpurun_out/par.log; exit 1; }
tail -5 gpurun_out/par.log
