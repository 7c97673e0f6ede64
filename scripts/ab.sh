#!/bin/bash
# A/B on the GPU box (one parametrised driver; the 17 one-off exp_*.sh of round 3 are gone):
#   ab.sh [-c CONFIG] [-k "<pytest -k expression>"] ENV1=.. ENV2=.. -- ENV3=.. -- ...
# Each env set (separated by --; "A=0" = the default build) gets a one-pair-in-flight line (stage times) and a pipelined line
# (throughput) of `bench.py --config CONFIG`; -k first runs the matching GPU tests.  Output: one summary line per env set.
cfg=2; kexpr=""
while getopts "c:k:" o; do case $o in c) cfg=$OPTARG;; k) kexpr=$OPTARG;; esac; done; shift $((OPTIND - 1))
cd ${GRAFT_REPO_ROOT:-.}
[ -n "$kexpr" ] && timeout 1500 python -m pytest tests -m gpu -x -q -k "$kexpr" 2>&1 | tail -3
common="--config $cfg --no-cpu --no-host-arrays --no-other-configs"
run() {
  rm -f profiles/bench_last_full.json /tmp/ab1.json /tmp/ab4.json; env "$@" python bench.py $common --streams 1 --batch 4 --steps 3 --warmup 1 > /tmp/ab1.out 2>&1; cp profiles/bench_last_full.json /tmp/ab1.json; rm -f profiles/bench_last_full.json
  env "$@" python bench.py $common --steps 6 --warmup 2 > /tmp/ab4.out 2>&1; cp profiles/bench_last_full.json /tmp/ab4.json
  python - "$*" <<PY
import json, sys
try:
    d = json.load(open("/tmp/ab1.json")); d4 = json.load(open("/tmp/ab4.json"))
    st = d["single_pair"]["stage_ms"]
    print(sys.argv[1], "| 1 in flight %.1f/s, one pair %.3f ms | pipelined %.1f/s (bitwise %s) |" % (d["value"], d["single_pair"]["ms"], d4["value"], d4["post_check"]["bitwise_equal"]),
          {k: round(v, 3) for k, v in st.items()})
except Exception as e:
    print(sys.argv[1], "FAILED", e, open("/tmp/ab1.out").read()[-800:], open("/tmp/ab4.out").read()[-800:])
PY
}
cur=()
for a in "$@"; do
  if [ "$a" == "--" ]; then run "${cur[@]}"; cur=(); else cur+=("$a"); fi
done
[ ${#cur[@]} -gt 0 ] && run "${cur[@]}"
