#!/bin/bash
# A/B of alternative builds of the library: ab_libs.sh libA.so libB.so ...   (single-stream stage times + pipelined throughput)
cp sfft_amd/libsfft_amd.so /tmp/lib_keep.so
for L in "$@"; do
  cp sfft_amd/$L sfft_amd/libsfft_amd.so
  python bench.py --streams 1 --batch 8 --steps 5 --warmup 2 --no-cpu --no-host-arrays 2>&1 | tail -1 > /tmp/o.json; python bench.py --steps 10 --warmup 2 --no-cpu --no-host-arrays 2>&1 | tail -1 > /tmp/o4.json
  python - "$L" <<PY
import json,sys
d=json.load(open("/tmp/o.json")); d4=json.load(open("/tmp/o4.json"))
st=d["single_pair"]["stage_ms"]
print(sys.argv[1], "| single %.3f ms | 4 streams %.1f/s |" % (d["single_pair"]["ms"], d4["value"]), {k: round(v,3) for k,v in st.items()}, "post", d4["post_check"]["bitwise_equal"])
PY
done
cp /tmp/lib_keep.so sfft_amd/libsfft_amd.so
