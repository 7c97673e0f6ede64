# A/B of the 6144- / 9216-point row and column kernels of fft_r24.hpp against the generic passes (configs 3 and 5)
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "forward_spectrum or r24 or config5 or config3 or four_step" 2>&1 | tail -3
run() { python bench.py --config $1 --steps 4 --warmup 2 --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); sp=d[\"single_pair\"]; print(d[\"value\"], sp[\"ms\"], {k:round(v,3) for k,v in sp[\"stage_ms\"].items()})"; }
for c in 3 5; do
echo config $c default; run $c
echo config $c generic rows; SFFT_NO_ROWS_R24=1 run $c
done
