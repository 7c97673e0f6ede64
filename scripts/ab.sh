#!/bin/bash
# A/B on the GPU box: ab.sh "<pytest -k expression>" ENV1=.. -- ENV2=.. ;  each env set gets a single-stream and a pipelined bench line
kexpr="$1"; shift
[ -n "$kexpr" ] && timeout 1200 python -m pytest tests -m gpu -x -q -k "$kexpr" 2>&1 | tail -3
run() { env "$@" python bench.py --streams 1 --batch 8 --steps 5 --warmup 2 --no-cpu --no-host-arrays 2>&1 | tail -1 > /tmp/o.json; env "$@" python bench.py --steps 10 --warmup 2 --no-cpu --no-host-arrays 2>&1 | tail -1 > /tmp/o4.json
python - "$*" <<PY
import json,sys
d=json.load(open("/tmp/o.json")); d4=json.load(open("/tmp/o4.json"))
st=d["single_pair"]["stage_ms"]
print(sys.argv[1], "| 1 stream %.1f/s single %.3f ms | 4 streams %.1f/s |" % (d["value"], d["single_pair"]["ms"], d4["value"]), {k: round(v,3) for k,v in st.items()}, "roof %.3f" % d["roofline_greek"]["frac"])
PY
}
cur=()
for a in "$@"; do
  if [ "$a" == "--" ]; then run "${cur[@]}"; cur=(); else cur+=("$a"); fi
done
run "${cur[@]}"
