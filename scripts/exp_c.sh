#!/bin/bash
# experiment: row chunks of the Omega launch at configs 3 / 5 (L2 sharing vs partial sums), and S = 2 / 3 with the single greek_g2 launch at config 2
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
one() { # cfg env...
  c=$1; shift
  env "$@" python bench.py --config $c --streams 1 --steps 3 --warmup 2 --no-cpu --no-host-arrays --no-other-configs 2>&1 | tail -1 > /tmp/o.json
  python - "$c $*" <<PY
import json,sys
d=json.load(open("/tmp/o.json"))
st=d["single_pair"]["stage_ms"]
print(sys.argv[1], "| %.2f/s single %.2f ms |" % (d["value"], d["single_pair"]["ms"]), {k: round(v,3) for k,v in st.items()})
PY
}
{
one 3 A=0; one 3 SFFT_G1_S=8; one 3 SFFT_G1_S=16; one 3 SFFT_G1_S=2
one 5 A=0; one 5 SFFT_G1_S=16; one 5 SFFT_G1_S=4
bash scripts/ab.sh "" SFFT_G1_BALANCE=0 -- SFFT_G1_S=2 -- SFFT_G1_S=3 -- SFFT_G1_RPC=1824
} 2>&1 | tee gpurun_out/exp_c.log
