#!/bin/bash
# experiment: configs 3 / 5 at their new pairs-in-flight with the Omega launch as one wave per SIMD (QUAD) and other switches
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { c=$1; shift
  env "$@" python bench.py --config $c --steps 4 --warmup 2 --no-cpu --no-host-arrays --no-other-configs 2>&1 | tail -1 > /tmp/o.json
  python - "cfg $c $*" <<PY
import json,sys
try:
    d=json.load(open("/tmp/o.json")); print(sys.argv[1], "| %.2f pairs/s | post" % d["value"], d["post_check"]["bitwise_equal"])
except Exception as e: print(sys.argv[1], "FAILED", e, open("/tmp/o.json").read()[-600:])
PY
}
{
run 3 A=0; run 3 SFFT_G1_QUAD=1; run 3 SFFT_INV_R24=1; run 3 SFFT_CHOL_LA=1
run 5 A=0; run 5 SFFT_G1_QUAD=1; run 5 SFFT_INV_R24=1; run 5 SFFT_CHOL_LA=1; run 5 SFFT_G1_S=16
run 3 A=0; run 5 A=0
} 2>&1 | tee gpurun_out/exp_y.log
