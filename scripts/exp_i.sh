#!/bin/bash
# experiment: one wave per SIMD with the loads of 2 x 4 steps in flight (twice the bytes in flight per SIMD of the default)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { L=$1; shift
  env SFFT_AMD_LIB=$PWD/sfft_amd/$L "$@" python bench.py --streams 1 --batch 8 --steps 5 --warmup 2 --no-cpu --no-host-arrays --no-other-configs 2>&1 | tail -1 > /tmp/o.json
  env SFFT_AMD_LIB=$PWD/sfft_amd/$L "$@" python bench.py --steps 10 --warmup 2 --no-cpu --no-host-arrays --no-other-configs 2>&1 | tail -1 > /tmp/o4.json
  python - "$L $*" <<PY
import json,sys
d=json.load(open("/tmp/o.json")); d4=json.load(open("/tmp/o4.json"))
st=d["single_pair"]["stage_ms"]
print(sys.argv[1], "| single %.3f ms | 4 streams %.1f/s | g1 %.3f g2 %.3f dit %s" % (d["single_pair"]["ms"], d4["value"], st["greek_g1"], st["greek_g2"], d["roofline_greek"].get("decimated")), "post", d4["post_check"]["bitwise_equal"])
PY
}
{
run libsfft_amd.so A=0
for L in lib_w1b4.so lib_w1b2.so; do run $L A=0; run $L SFFT_G1_S=2; run $L SFFT_G1_S=8; done
} 2>&1 | tee gpurun_out/exp_i.log
