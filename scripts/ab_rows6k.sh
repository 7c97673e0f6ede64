# A/B of the 6144-point row / column kernels against the generic passes (config 3)
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "bspline or config3 or generic or forward_spectrum" 2>&1 | tail -3
run() { python bench.py --config 3 --steps 4 --warmup 2 --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); sp=d[\"single_pair\"]; print(d[\"value\"], sp[\"ms\"], {k:round(v,3) for k,v in sp[\"stage_ms\"].items()})"; }
echo default; run
echo norows6k; SFFT_NO_ROWS6K=1 run
