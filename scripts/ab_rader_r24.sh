# A/B of the register-resident Rader 577 sub-transform (config 5's 9232-point column axis)
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "forward_spectrum or four_step or config5" 2>&1 | tail -3
run() { python bench.py --config 5 --steps 4 --warmup 2 --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); sp=d[\"single_pair\"]; print(d[\"value\"], sp[\"ms\"], {k:round(v,3) for k,v in sp[\"stage_ms\"].items()})"; }
echo default; run
echo lds rader; SFFT_NO_RADER_R24=1 run
