#!/bin/bash
# experiment: greek_g1_mfma4w (workgroups of eight waves, planes shared through LDS) against the one-wave-per-group launch
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
one() { c=$1; shift
  env "$@" python bench.py --config $c --streams 1 --steps 3 --warmup 2 --no-cpu --no-host-arrays --no-other-configs 2>&1 | tail -1 > /tmp/o.json
  python - "$c $*" <<PY
import json,sys
try:
    d=json.load(open("/tmp/o.json")); st=d["single_pair"]["stage_ms"]
    print(sys.argv[1], "| %.2f/s single %.2f ms |" % (d["value"], d["single_pair"]["ms"]), {k: round(v,3) for k,v in st.items()}, "post", d["post_check"]["bitwise_equal"])
except Exception as e: print(sys.argv[1], "FAILED", e, open("/tmp/o.json").read()[-600:])
PY
}
{
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or full_size or lhmat or system or config3 or config5" 2>&1 | tail -8
bash scripts/ab.sh "" SFFT_G1_WG=0 -- SFFT_G1_WG=1 -- SFFT_G1_WG=1 SFFT_G1_S=2
one 3 SFFT_G1_WG=0; one 3 SFFT_G1_WG=1; one 5 SFFT_G1_WG=0; one 5 SFFT_G1_WG=1
} 2>&1 | tee gpurun_out/exp_g.log
