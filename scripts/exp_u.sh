#!/bin/bash
# experiment: odd Fij -- the last Theta pass as a slot beside the lone diagonal pass (default) vs a vector launch of its own (SFFT_THETA_SLOTS=0)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
one() { c=$1; shift
  env "$@" python bench.py --config $c --streams 1 --steps 3 --warmup 2 --no-cpu --no-host-arrays --no-other-configs 2>&1 | tail -1 > /tmp/o.json
  env "$@" python bench.py --config $c --steps 5 --warmup 2 --no-cpu --no-host-arrays --no-other-configs 2>&1 | tail -1 > /tmp/o2.json
  python - "$c $*" <<PY
import json,sys
try:
    d=json.load(open("/tmp/o.json")); st=d["single_pair"]["stage_ms"]; d2=json.load(open("/tmp/o2.json"))
    print(sys.argv[1], "| default streams %.2f/s | single %.2f ms |" % (d2["value"], d["single_pair"]["ms"]), {k: round(v,3) for k,v in st.items()}, "post", d2["post_check"]["bitwise_equal"])
except Exception as e: print(sys.argv[1], "FAILED", e, open("/tmp/o.json").read()[-800:])
PY
}
{
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "config3 or bspline or bs_ or omega_launch_variants" 2>&1 | tail -5
one 3 SFFT_THETA_SLOTS=0; one 3 SFFT_THETA_SLOTS=1; one 3 SFFT_THETA_SLOTS=0; one 3 SFFT_THETA_SLOTS=1
} 2>&1 | tee gpurun_out/exp_u.log
