#!/bin/bash
# experiment: register-resident inverse row pass for 6144- / 9216-point rows vs the generic LDS-resident one (SFFT_NO_INV_R24=1); 9216 points behind SFFT_INV_R24_9216=1
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
one() { c=$1; shift
  env "$@" python bench.py --config $c --streams 1 --steps 3 --warmup 2 --no-cpu --no-host-arrays --no-other-configs 2>&1 | tail -1 > /tmp/o.json
  env "$@" python bench.py --config $c --steps 5 --warmup 2 --no-cpu --no-host-arrays --no-other-configs 2>&1 | tail -1 > /tmp/o2.json
  python - "$c $*" <<PY
import json,sys
try:
    d=json.load(open("/tmp/o.json")); st=d["single_pair"]["stage_ms"]; d2=json.load(open("/tmp/o2.json"))
    print(sys.argv[1], "| default streams %.2f/s | single %.2f ms | inverse %.3f" % (d2["value"], d["single_pair"]["ms"], st["inverse"]), "post", d2["post_check"]["bitwise_equal"])
except Exception as e: print(sys.argv[1], "FAILED", e, open("/tmp/o.json").read()[-800:])
PY
}
{
SFFT_INV_R24_9216=1 timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "config3 or config5 or strip_matches or r24 or large_shapes" 2>&1 | tail -3
one 3 SFFT_NO_INV_R24=1; one 3 A=0; one 3 SFFT_NO_INV_R24=1; one 3 A=0; one 5 A=0; one 5 SFFT_INV_R24_9216=1; one 5 A=0; one 5 SFFT_INV_R24_9216=1
} 2>&1 | tee gpurun_out/exp_w.log
