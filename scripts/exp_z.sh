#!/bin/bash
# experiment: config 5's tap walk (KerHW 12, order 3) two source rows per table read (vconv_mixed2, SFFT_VCONV2_W12=1) vs one (vconv_mixed)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
one() { c=$1; shift
  env "$@" python bench.py --config $c --streams 1 --batch 2 --steps 3 --warmup 2 --no-cpu --no-host-arrays --no-other-configs 2>&1 | tail -1 > /tmp/o.json
  env "$@" python bench.py --config $c --steps 4 --warmup 2 --no-cpu --no-host-arrays --no-other-configs 2>&1 | tail -1 > /tmp/o2.json
  python - "$c $*" <<PY
import json,sys
try:
    d=json.load(open("/tmp/o.json")); st=d["single_pair"]["stage_ms"]; d2=json.load(open("/tmp/o2.json"))
    print(sys.argv[1], "| %.2f pairs/s | single %.2f ms | construct %.3f" % (d2["value"], d["single_pair"]["ms"], st["construct"]), "post", d2["post_check"]["bitwise_equal"])
except Exception as e: print(sys.argv[1], "FAILED", e, open("/tmp/o.json").read()[-800:])
PY
}
{
SFFT_VCONV2_W12=1 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "config5 or strip_matches or mixed_domain or omega_launch_variants" 2>&1 | tail -3
one 5 A=0; one 5 SFFT_VCONV2_W12=1; one 5 A=0; one 5 SFFT_VCONV2_W12=1
} 2>&1 | tee gpurun_out/exp_z.log
