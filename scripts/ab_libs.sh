#!/bin/bash
# A/B of alternative builds of the library: ab_libs.sh libA.so libB.so ...   (single-stream stage times + pipelined throughput)
for L in "$@"; do
  cp sfft_amd/$L sfft_amd/libsfft_amd.so
  python bench.py --streams 1 --batch 8 --steps 5 --warmup 2 --no-cpu --no-host-arrays 2>&1 | tail -1 > /tmp/o.json; python bench.py --steps 10 --warmup 2 --no-cpu --no-host-arrays 2>&1 | tail -1 > /tmp/o4.json
  python - "$L" <<PY
import json,sys
d=json.load(open("/tmp/o.json")); d4=json.load(open("/tmp/o4.json"))
st=d["single_pair"]["stage_ms"]
print(sys.argv[1], "| single %.3f ms | 4 streams %.1f/s | greek_g1 %.3f" % (d["single_pair"]["ms"], d4["value"], st["greek_g1"]), "post", d4["post_check"]["bitwise_equal"])
PY
done
