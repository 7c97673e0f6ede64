#!/bin/bash
# experiment: the solver's 16 x 16 x 4 matrix products as four 4 x 4 x 4 ones (SFFT_CHOL_M4=1, lib_m4.so) against the default build
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
one() { L=$1; c=$2
  env SFFT_AMD_LIB=$PWD/sfft_amd/$L python bench.py --config $c --streams 1 --steps 3 --warmup 2 --no-cpu --no-host-arrays --no-other-configs 2>&1 | tail -1 > /tmp/o.json
  python - "$L cfg$c" <<PY
import json,sys
d=json.load(open("/tmp/o.json")); st=d["single_pair"]["stage_ms"]
print(sys.argv[1], "| %.2f/s single %.2f ms | solve %.3f" % (d["value"], d["single_pair"]["ms"], st["solve"]), "post", d["post_check"]["bitwise_equal"])
PY
}
{
SFFT_AMD_LIB=$PWD/sfft_amd/lib_m4.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or solver or outer_blocked or lu_redo or graph or full_size or config3 or config5" 2>&1 | tail -3
bash scripts/ab_libs.sh lib_base.so lib_m4.so lib_base.so lib_m4.so
for c in 3 5; do one lib_base.so $c; one lib_m4.so $c; done
} 2>&1 | tee gpurun_out/exp_m4.log
