#!/bin/bash
# experiment: pairs in flight for configs 3 and 5
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { c=$1; st=$2; b=$3
  python bench.py --config $c --streams $st --batch $b --steps 4 --warmup 2 --no-cpu --no-host-arrays --no-other-configs 2>&1 | tail -1 > /tmp/o.json
  python - "cfg $c streams $st batch $b" <<PY
import json,sys
try:
    d=json.load(open("/tmp/o.json")); print(sys.argv[1], "| %.2f pairs/s | post" % d["value"], d["post_check"]["bitwise_equal"])
except Exception as e: print(sys.argv[1], "FAILED", e, open("/tmp/o.json").read()[-600:])
PY
}
{
run 5 1 2; run 5 2 4; run 5 3 6; run 3 2 4; run 3 3 6; run 3 4 8; run 5 1 2; run 5 2 4
} 2>&1 | tee gpurun_out/exp_x.log
