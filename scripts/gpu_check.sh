mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 2>&1 | tail -15 > gpurun_out/t2.log; tail -5 gpurun_out/t2.log
for v in 0 1 2; do SFFT_G1_VARIANT=$v timeout 300 python bench.py --steps 5 --warmup 2 --cpu-sample 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('variant $v', round(d['value'],2), {k:round(x,3) for k,x in d['stage_ms'].items()})"; done
