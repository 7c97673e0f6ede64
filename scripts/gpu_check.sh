mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 2>&1 | tail -15 > gpurun_out/t2.log; tail -4 gpurun_out/t2.log
python bench.py --streams 1 --steps 8 --warmup 2 --cpu-sample 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('streams1', round(d['value'],2), {k:round(x,3) for k,x in d['stage_ms'].items()})"
python bench.py --steps 8 --warmup 2 --cpu-sample 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default', round(d['value'],2), 'roofline', d['roofline']['kernel'], round(d['roofline']['frac'],3), '| hbm', d['roofline_hbm']['kernel'], round(d['roofline_hbm']['frac'],3))"
