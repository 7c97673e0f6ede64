#!/bin/bash
# A/B of the dataflow Cholesky: solver-related GPU tests, then single-stream and pipelined bench lines for a few settings
timeout 900 python -m pytest tests -m gpu -x -q -k "linear_system or lu or solver or pccp or stationary or config3 or poison or error or gss or full_size" 2>&1 | tail -4
run() { env "$@" python bench.py --streams 1 --batch 8 --steps 5 --warmup 2 --no-cpu --no-host-arrays 2>&1 | tail -1 > /tmp/o.json; env "$@" python bench.py --steps 10 --warmup 2 --no-cpu --no-host-arrays 2>&1 | tail -1 > /tmp/o4.json
python - "$*" <<PY
import json,sys
d=json.load(open("/tmp/o.json")); d4=json.load(open("/tmp/o4.json"))
print(sys.argv[1], "| 1 stream %.1f/s single %.3f ms solve %.3f | 4 streams %.1f/s" % (d["value"], d["single_pair"]["ms"], d["single_pair"]["stage_ms"]["solve"], d4["value"]))
PY
}
run SFFT_CHOL_DF=0
run SFFT_CHOL_DF=1
run SFFT_CHOL_DF_WG=32
run SFFT_CHOL_DF_WG=128
