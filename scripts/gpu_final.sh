set -e
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^$" > gpurun_out/final_gpu.log || { tail -40 gpurun_out/final_gpu.log; exit 1; }
tail -3 gpurun_out/final_gpu.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout -k 10 600 python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err; echo "bench rc=$?"
python -c "
import json
j=json.loads(open('gpurun_out/final_bench.json').read().strip().splitlines()[-1])
print(j['metric'], j['value'], j['ms_per_step'], j['roofline']['frac'])
"
timeout -k 10 300 python bench.py --degree 32 --colloc 64 --elements 100000 --domain narrow --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('deg32', j['value'], j['ms_per_step'], j['roofline']['frac'])"
